// Shared by the translation units of libprotnote_hip.so (not part of the C ABI).
#pragma once

namespace pn {
// printf-style: stores the message pn_last_error() returns (thread-local) and returns 1.
__attribute__((visibility("hidden"))) int fail_msg(const char* fmt, ...);
}  // namespace pn
