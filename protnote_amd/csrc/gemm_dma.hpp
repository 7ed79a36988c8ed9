// f32-MFMA "NT" GEMM for gfx950 with LDS-DMA operand staging (global_load_lds_dwordx4): the 256x256-tile kernel of the
// pair-grid GEMMs, C[M,N] = gen(A)[M,K] * W[N,K]^T, same operand generators / epilogues / tile order as
// gemm_engine.hpp.  What changes is how operands reach the LDS:
//   * W (always a plain matrix) goes global -> LDS directly, no VGPR round trip, no ds_write pass;
//   * A_PLAIN (the dh = dz * W GEMMs of the backward) stages A the same way: the main loop is then nothing but
//     fragment reads + MFMAs, one barrier per slab;
//   * generated A operands (relu(s*z+t), relu(A'[i]+B'[j])) keep the register path, written under the second half
//     of the slab's MFMAs.
// An LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB of CONTIGUOUS LDS, so rows cannot be padded: the LDS
// image is [row][32 floats] (128 B) with the 16-byte granule g of row r stored at position g ^ ((r >> 1) & 7).  The
// swizzle is applied on the per-lane SOURCE address of the DMA (and on the ds_write of the register path) and undone by
// the fragment reads; every 16-lane group of a ds_read_b128 (rows r..r+15 at one k-granule) then covers 16 distinct
// 16-byte slots - conflict-free, like the padded image of gemm_engine.hpp.
// Pipeline per slab s (BK = 32, 128 MFMAs per wave = ~16k cycles per SIMD): issue the DMA (and the register loads) of
// slab s+1 into the other buffer, compute slab s, then s_waitcnt vmcnt(0) + s_barrier: every load has a whole slab
// (or, for the register operand, half of one) to land.  The DMA is issued from inline asm (the compiler does not know
// about it, so it cannot put a conservative vmcnt(0) in front of the fragment reads); its completion is ordered for the
// readers by the explicit vmcnt(0) of the issuing wave followed by the barrier (MI355X_MICROARCH.md, LDS-DMA item 7).
// Restrictions: one K segment, K % 32 == 0, N % 256 == 0 (the pair-grid shapes).
#pragma once
#include "gemm_engine.hpp"

namespace pn {

// one LDS-DMA wave-instruction: lane l copies 16 bytes from its own global address to LDS[lds_base + 16 l]
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_base_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_uniform)
      : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// one LDS-DMA wave-instruction in SGPR-base form: lane l copies 16 bytes from sbase + voff[l] to LDS[lds_base + 16 l].
// The uniform part of the address (tile origin + slab offset) lives in scalar registers and is advanced by the scalar
// unit; the per-lane byte offset is a loop-invariant VGPR - no vector instruction is spent on addresses in the slab loop.
__device__ __forceinline__ void glds16s(const float* sbase_uniform, unsigned voff_bytes, unsigned lds_base_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(sbase_uniform), "s"(lds_base_uniform)
      : "memory");
}
// global load of 16 bytes from a uniform base + a 32-bit per-lane byte offset (selects the SGPR-base addressing mode)
__device__ __forceinline__ float4 ld4_so(const float* sbase_uniform, unsigned voff_bytes) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sbase_uniform) + voff_bytes);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// The same load issued from inline asm in its SGPR-base form (left to itself the optimiser folds the loop-invariant
// per-lane offset into a 64-bit per-lane pointer and adds the slab offset with one vector instruction per load).  The
// compiler does not track it: the caller waits with wait_loads() - which takes the destination registers as operands, so
// nothing can consume them earlier - before the first use.
__device__ __forceinline__ void gload4_s(f32x4& dst, const float* sbase_uniform, unsigned voff_bytes) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase_uniform) : "memory");
}
// s_waitcnt vmcnt(VM) with the awaited registers as operands: no consumer can be scheduled above it
template <int VM>
__device__ __forceinline__ void wait_vm6(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(VM) : "memory");
}
template <int VM>
__device__ __forceinline__ void wait_vm8(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f, f32x4& g, f32x4& h) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
               : "n"(VM)
               : "memory");
}
__device__ __forceinline__ float relu_raw(float x) {  // v_max_f32 without the canonicalising self-max the compiler
  float r;                                            // puts in front of fmaxf on values it did not compute itself
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
#define PN_LDS __attribute__((address_space(3)))
// packed f32 arithmetic, one issue for two values (the compiler mostly scalarises <2 x float> operations)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float4 lds_read4(unsigned addr) {
  const f32x4 v = *reinterpret_cast<const PN_LDS f32x4*>(addr);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_write4(unsigned addr, float4 v) {
  *reinterpret_cast<PN_LDS f32x4*>(addr) = f32x4{v.x, v.y, v.z, v.w};
}

// Every vector instruction that is not an MFMA costs matrix-pipe time on this chip: v_mfma_f32_32x32x2_f32 occupies the
// SIMD for 64 cycles and any other VALU instruction of either resident wave delays the next MFMA issue - measured with
// dummy v_mov's in the all-DMA loop (tools/gemm_latency_probe.py, -DPN_VALU_PROBE): 32 extra VALU instructions per slab
// and wave cost 2.3 %, 64 cost 3.8 % (and memory latency costs nothing: operands served from one cache line run no
// faster).  So the slab loop is written to spend vector instructions on nothing but the operand transform itself:
//   * addresses: uniform part in SGPRs (scalar adds), per-lane part loop-invariant (glds16s / ld4_so);
//   * the loop is unrolled over the two LDS buffers, so buffer and tile offsets fold into the ds_read / ds_write
//     immediate offsets of precomputed per-lane byte addresses (LDS: A0 | A1 | B0 | B1, 32 KiB each);
//   * the BatchNorm affine of the generated operand uses packed f32 FMAs (v_pk_fma_f32: two lanes' worth per issue).
// all-DMA loop 18 -> 0, relu(s*z+t) loop 68 -> ~25, pair-sum loop 82 -> ~25 vector instructions per slab and wave.
// (Measured and dropped, DESIGN.md 4.0: persistent workgroups that pipeline through the tile boundary - 149.8 -> 147.3
// TFLOP/s, loads and stores share the in-order vmcnt; issuing all of a slab's DMA at its top, or the weight tile after the
// second k-step only.)
template <int AK, int EK, bool DROP = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_dma_kernel(const GemmParams p) {
  static_assert(!DROP || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "dropout applies to the hidden activations");
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 4, BK = 32;
  constexpr int BM = 256, BN = 256;
  constexpr bool A_DMA = (AK == A_PLAIN);
  static_assert(AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "operand kind not built for the DMA kernel");
  constexpr int TILE = BM * BK;             // floats per operand buffer (32 KiB)
  constexpr unsigned TILEB = TILE * 4u;     // LDS bytes: A buffer c at c * TILEB, B buffer c at (2 + c) * TILEB

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  int tile_m, tile_n;
  if (!tile_coords<BM, BN>(p, tile_m, tile_n)) return;
  const int row0 = tile_m * BM;
  const int col0 = tile_n * BN;
  const int nslab = p.Kseg / BK;

  // ---- DMA sources: wave w, instruction q covers tile rows 8 (4 w + q) .. + 7; lane l: row + l / 8, LDS granule
  //      position l % 8 holds source granule (l % 8) ^ ((row >> 1) & 7).  Byte offsets relative to the tile origin.
  const unsigned lds0 = lds_addr(smem);
  const float* w_tile = p.W + (long)col0 * p.ldw;
  const float* a_tile = p.A + (long)row0 * p.lda;
  unsigned boff[4], aoff_dma[4];  // (A_PAIRSUM_RELU addresses its two tables from their origins)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    boff[q] = (unsigned)((long)r * p.ldw + 4 * g) * 4u;
    int ra_ = row0 + r;
    if (ra_ > p.M - 1) ra_ = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    aoff_dma[q] = (unsigned)((long)(ra_ - row0) * p.lda + 4 * g) * 4u;
  }
  auto issue_b = [&](int s, auto buf_c, int q0 = 0, int q1 = 4) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = w_tile + s * BK;
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_a = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = a_tile + s * BK;
    const unsigned base = lds0 + BUF * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(src, aoff_dma[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };

  // ---- register path of a generated A operand: thread = (row r_in + 64 q, granule kv), 4 rows per thread
  constexpr int KV = 8, RPP = 512 / KV, NQA = BM / RPP;
  const int kv = tid % KV;
  const int r_in = tid / KV;
  unsigned aoff[NQA], aoff2[NQA];  // per-lane byte offsets of the operand rows (from a_tile, or from the table origins)
  unsigned awr[NQA];               // LDS byte offset of the quad inside an A buffer (swizzled)
  uint32_t a_key[NQA];             // DROP: per-row key of the dropout hash (gemm_engine.hpp)
  int a_col = 0;
  if constexpr (!A_DMA) {
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      const int rl = r_in + q * RPP;
      int r = row0 + rl;
      if (r > p.M - 1) r = p.M - 1;
      a_key[q] = DROP ? drop_rowkey(p.drop_seed, (uint32_t)r) : 0u;
      if constexpr (AK == A_PAIRSUM_RELU) {
        const int j = r / p.pairB;
        const int i = r - j * p.pairB;
        aoff[q] = (unsigned)((long)i * p.lda + 4 * kv) * 4u;
        aoff2[q] = (unsigned)((long)j * p.lda2 + 4 * kv) * 4u;  // the launcher checks that the table fits 32 bits
      } else {
        aoff[q] = (unsigned)((long)(r - row0) * p.lda + 4 * kv) * 4u;
        aoff2[q] = 0u;
      }
      awr[q] = lds0 + (unsigned)(rl * BK + 4 * (kv ^ ((rl >> 1) & 7))) * 4u;
      asm volatile("" : "+v"(awr[q]));  // keep it a register: the buffer offset then folds into the ds_write immediate
    }
  }
  const unsigned st_off = 16u * kv;  // scale / shift quad of this thread
  f32x4 ra[NQA], ra2[NQA];
  f32x4 rsc = {0.f, 0.f, 0.f, 0.f}, rsh = rsc;
  auto fetch_a = [&](int s) {
    const int c = s * BK;
    if constexpr (DROP) a_col = c + 4 * kv;
    const float* src = (AK == A_PAIRSUM_RELU ? p.A : a_tile) + c;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      gload4_s(ra[q], src, aoff[q]);
      if constexpr (AK == A_PAIRSUM_RELU) gload4_s(ra2[q], p.A2 + c, aoff2[q]);
    }
    if constexpr (AK == A_AFFINE_RELU) {
      gload4_s(rsc, p.a_scale + c, st_off);
      gload4_s(rsh, p.a_shift + c, st_off);
    }
  };
  // wait for the register operand (and, vmcnt being in order, the DMA issued before it): the destination registers are
  // operands of the wait, so no consumer can be scheduled above it
  auto pin_a = [&](auto vm_c) {
    constexpr int PIN_VMCNT = decltype(vm_c)::value;
    static_assert(NQA == 4, "operand list below");
    // (the four W-tile DMAs issued after the first k-step are YOUNGER than the register loads - the counter is in order,
    //  "at most 4 outstanding" is the register operand complete with the DMA still in flight)
    if constexpr (AK == A_PAIRSUM_RELU) wait_vm8<PIN_VMCNT>(ra[0], ra[1], ra[2], ra[3], ra2[0], ra2[1], ra2[2], ra2[3]);
    else wait_vm6<PIN_VMCNT>(ra[0], ra[1], ra[2], ra[3], rsc, rsh);
  };
  auto commit_a = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      float4 v;
      if constexpr (AK == A_AFFINE_RELU) {  // relu(s * z + t): two packed FMAs + four max (same fma, same rounding)
        const f32x2 lo = pk_fma(ra[q].xy, rsc.xy, rsh.xy);
        const f32x2 hi = pk_fma(ra[q].zw, rsc.zw, rsh.zw);
        v = make_float4(relu_raw(lo.x), relu_raw(lo.y), relu_raw(hi.x), relu_raw(hi.y));
      } else {
        const f32x2 lo = pk_add(ra[q].xy, ra2[q].xy);
        const f32x2 hi = pk_add(ra[q].zw, ra2[q].zw);
        v = make_float4(relu_raw(lo.x), relu_raw(lo.y), relu_raw(hi.x), relu_raw(hi.y));
      }
      if constexpr (DROP) v = drop4(v, a_key[q], (uint32_t)a_col, p.drop_thresh, p.drop_scale);
      lds_write4(awr[q] + BUF * TILEB, v);
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 32 of its wave tile and k-granule 2 kk + l / 32 of k-step kk, stored at
  // granule position (2 kk + l / 32) ^ ((row >> 1) & 7); (row >> 1) & 7 == (l >> 1) & 7 for every tile of the wave.
  // One precomputed byte address per k-step and operand; buffer and tile offsets are immediates.
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int sw = (lane >> 1) & 7;
  unsigned fa_addr[4], fb_addr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int fo = 4 * ((2 * kk + fh) ^ sw);
    fa_addr[kk] = lds0 + (unsigned)((wm * WM * 32 + frow) * BK + fo) * 4u;
    fb_addr[kk] = lds0 + 2u * TILEB + (unsigned)((wn * WN * 32 + frow) * BK + fo) * 4u;
    asm volatile("" : "+v"(fa_addr[kk]), "+v"(fb_addr[kk]));
  }
  auto read_frag = [&](auto buf_c, auto kk_c, float4 (&a)[WM], float4 (&b)[WN]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = lds_read4(fa_addr[KK] + (BUF * TILEB + i * 32 * BK * 4));
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = lds_read4(fb_addr[KK] + (BUF * TILEB + j * 32 * BK * 4));
  };
  auto mma = [&](const float4 (&a)[WM], const float4 (&b)[WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
      }
  };

  auto mma_rows = [&](const float4 (&a)[WM], const float4 (&b)[WN], auto i_c) {  // one 32-row strip of the wave tile
    constexpr int i = decltype(i_c)::value;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    }
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I2 = integral_constant<int, 2>;
  using I3 = integral_constant<int, 3>;
  // ---- prologue: slab 0 into buffer 0
  issue_b(0, I0{});
  if constexpr (A_DMA) {
    issue_a(0, I0{});
  } else {
    fetch_a(0);
    pin_a(I0{});
    commit_a(I0{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One slab out of buffer CUR.  Rotated loop: the last k-step's 32 MFMAs of slab s are issued AFTER the barrier that
  // publishes slab s+1, behind the fragment reads of slab s+1's first k-step - the matrix pipe has work while those reads
  // are in flight (the fragment registers alternate between two sets).  The slab after the last re-stages the last one
  // into the idle buffer (branch-free: a conditional prefetch makes hipcc wait where the loads are issued).  A generated
  // operand of slab s+1 is fetched at the slab top (half a slab to land), transformed and written under k-step 2's MFMAs.
  // Measured on M = 524288, K = 6144 (all-DMA loop, before the address work): 144.0 -> 145.1 TFLOP/s from the rotation;
  // without the DMA 150.4, without DMA and barrier 151.8 (MFMA + fragment reads); spreading the DMA instructions over the
  // k-steps 139.3; a static s_setprio for half of the workgroup 0 %.
  float4 fa[WM], fb[WN], ga[WM], gb[WN];
  auto slab = [&](int s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    const int nxt = s + 1 < nslab ? s + 1 : s;
    // The DMA / load issue of slab s+1 is spread over the slab instead of bursting at its top (where both waves of a SIMD
    // would sit in ~100 scalar + VMEM instructions with the matrix pipe idle): the operand that streams from HBM goes
    // first, the L2-resident weight tile after the first k-step - in the all-DMA kernels in two halves, after the first and
    // the second (148.5 -> 150.0 -> 150.5 TFLOP/s on the probe GEMM).  (Before the address work this spreading cost 4 %: the
    // per-instruction 64-bit vector adds landed between the MFMAs.)
    if constexpr (A_DMA) issue_a(nxt, N{});  // the other buffer was last read in slab s-1
    else fetch_a(nxt);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I1{}, ga, gb);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (A_DMA) issue_b(nxt, N{}, 0, 2);  // all-DMA loop: half of the weight tile here, half after k-step 1
    else issue_b(nxt, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I2{}, fa, fb);
    mma(ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (A_DMA) {
      issue_b(nxt, N{}, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    read_frag(C{}, I3{}, ga, gb);
    if constexpr (A_DMA) {
      mma(fa, fb);
    } else {  // the register operand is waited for as late as its transform + LDS write still fit under MFMAs: after the
              // first 16 of k-step 2's 32 MFMAs (three quarters of a slab to land)
      mma_rows(fa, fb, I0{});
      __builtin_amdgcn_sched_barrier(0);
      pin_a(integral_constant<int, 4>{});
      mma_rows(fa, fb, I1{});
      commit_a(N{});
    }
    __builtin_amdgcn_sched_barrier(0);
    // DMA of slab s+1 complete for this wave, its LDS writes and this wave's fragment reads drained; then the barrier
    // makes every wave's share visible (and frees buffer CUR for the DMA of slab s+2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frag(N{}, I0{}, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    mma(ga, gb);
  };
  read_frag(I0{}, I0{}, fa, fb);
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab(s, I0{});
    slab(s + 1, I1{});
  }
  if (s < nslab) slab(s, I0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

constexpr int GEMM_DMA_LDS_BYTES = 2 * 2 * 256 * 32 * (int)sizeof(float);  // 128 KiB

}  // namespace pn
