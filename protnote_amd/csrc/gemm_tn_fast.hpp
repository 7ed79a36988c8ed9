// f32 weight-gradient (TN) kernel of the pair head, specialised for the shapes that carry the train step:
//     dW[m][n] = sum_r dz[r][m] * h[r][n],   h = relu(s * z + t)  (TB_AFFINE_RELU)  or  relu(A'[r % B] + B'[r / B])  (TB_PAIRSUM_RELU)
// M, N multiples of 256, every row split a whole number of 32-row slabs, for the pair-sum kind B % 32 == 0 (a slab then
// lies inside one label: B'[j] is ONE row per slab and the A' rows are consecutive).  Same tiles, LDS images, product
// order and split-K partials as gemm_tn_kernel<TA_PLAIN, TB, BIG, ADMA> - results are bit-identical - but written, like
// gemm_nt_dma_kernel, so that the slab loop spends vector instructions on nothing but the operand transform: the generic
// kernel's row clamps, masks, 64-bit per-lane addresses and pair decode cost ~125-160 VALU instructions per slab and wave,
// and every one of them takes matrix-pipe time (gemm_dma.hpp).  Here: uniform address parts in SGPRs (row * ld, the pair
// decode, the slab offset), one loop-invariant per-lane byte offset per operand, loop unrolled over the two LDS buffers
// (A0 | A1 | B0 | B1, 32 KiB each: every ds offset is an immediate), packed f32 FMA / add: 24 VALU per slab and wave.
#pragma once
#include "gemm_dma.hpp"
#include "gemm_tn.hpp"

#ifndef PN_TN_SYNC_SLABS
#define PN_TN_SYNC_SLABS 8  // slabs per pacing checkpoint (power of two); measured 4 / 8 / 16: 0.54 / 0.62 / 0.76 TB per launch at 143.8 / 145.6 / 146.0 TFLOP/s
#endif
#ifndef PN_TN_WAIT_KK
#define PN_TN_WAIT_KK 14
#endif

namespace pn {

__device__ __forceinline__ float2 lds_read2(unsigned addr) {
  const f32x2 v = *reinterpret_cast<const PN_LDS f32x2*>(addr);
  return make_float2(v.x, v.y);
}
// 16-byte global load through a buffer descriptor: base (uniform, advanced by the scalar unit) in SGPRs, per-lane byte
// offset in a loop-invariant VGPR - no vector instruction for the address, and, unlike the inline-asm loads of
// gemm_dma.hpp, tracked by the compiler (it places the s_waitcnt itself, so the result may stay in flight across the
// loop edge).  Word 3 = 0x00020000: untyped 32-bit data, no swizzle (gfx90a / gfx94x / gfx950); 4 GB window from the base.
__device__ __forceinline__ f32x4 bload4(const float* sbase_uniform, unsigned voff_bytes) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sbase_uniform), 0, -1, 0x00020000);
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  const i32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void lds_write2(unsigned addr, float2 v) {
  *reinterpret_cast<PN_LDS f32x2*>(addr) = f32x2{v.x, v.y};
}

template <int I0_, int I1_, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0_ < I1_) {
    f(std::integral_constant<int, I0_>{});
    static_for<I0_ + 1, I1_>(f);
  }
}

// SYNC: the region task's workgroups pace each other (checkpoint below); compiled in only where it is used - the code in the
// slab loop costs 1-3 % even when it does nothing.
template <int TB, bool SYNC = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_fast_kernel(const TnParams p) {
  static_assert(TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU, "operand kind not built for the fast TN kernel");
  constexpr int BM = 256, BN = 256, BK = 32, NQ = 4;
  constexpr unsigned ROWB = 1024u;    // bytes of one tile row (256 floats)
  constexpr unsigned TILEB = 32768u;  // bytes of one operand buffer (32 rows)

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = p.N / BN, ntm = p.M / BM;
  int tile_m, tile_n;
  int split = blockIdx.y;
  int* sync_ctr = nullptr;  // this task's four arrival counters (whole-split tasks only: equal work for all 32 workgroups)
  if (p.task_ns > 0) {
    if (!tn_task_coords(p.task_ns, tile_m, tile_n, split)) return;
    const int T = ((int)blockIdx.x >> 8) * 8 + ((int)blockIdx.x & 7);
    if (SYNC && p.task_sync != nullptr && T < p.task_ns * 4) sync_ctr = p.task_sync + 4 * T;
  } else if (PN_XCD && (ntm % 2 == 0) && (ntn % 4 == 0)) {
    const int rm = ntm / 2, rn = ntn / 4;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 2) * rm + w / rn;
    tile_n = (xcd & 3) * rn + w % rn;
  } else {
    tile_n = blockIdx.x % ntn;
    tile_m = blockIdx.x / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long r_begin = (long)split * p.rows_per_split;
  long r_end = r_begin + p.rows_per_split;
  if (r_end > p.R) r_end = p.R;
  const int nsl = r_end > r_begin ? (int)((r_end - r_begin) / BK) : 0;  // whole slabs (launcher's precondition)
  const unsigned lds0 = lds_addr(smem);

  // ---- A (the materialised dz) by LDS-DMA: wave w stages tile rows w + 8 q of a slab, lane l the columns 4 l .. 4 l + 3
  const float* a_col = p.A + m0;
  const unsigned a_lane = 16u * lane;
  auto issue_a = [&](int t, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* row = a_col + (r_begin + (long)t * BK + wave) * p.lda;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      glds16s(row + (long)(8 * q) * p.lda, a_lane,
              __builtin_amdgcn_readfirstlane(lds0 + BUF * TILEB + (unsigned)(wave + 8 * q) * ROWB));
  };

  // ---- B through registers: thread (wave, lane) loads rows wave + 8 q, columns 4 lane .. + 3 of the slab
  const unsigned b_lane = (unsigned)((long)wave * p.ldb + 4 * lane) * 4u;
  const unsigned b2_lane = 16u * lane;
  f32x4 rb[NQ], rb2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs = {0.f, 0.f, 0.f, 0.f}, bt = bs;
  if constexpr (TB == TB_AFFINE_RELU) {
    const float4 s4 = ld4(p.b_s + n0 + 4 * lane), t4 = ld4(p.b_t + n0 + 4 * lane);
    bs = f32x4{s4.x, s4.y, s4.z, s4.w};
    bt = f32x4{t4.x, t4.y, t4.z, t4.w};
  }
  // pair decode of the slab to fetch next, carried in scalar registers: row r = j * B + i, a slab lies inside one label
  int pf_i = 0, pf_j = 0;
  if constexpr (TB == TB_PAIRSUM_RELU) {
    pf_j = (int)(r_begin / p.pairB);
    pf_i = (int)(r_begin - (long)pf_j * p.pairB);
  }
  int pf_t = 0;  // slab the carried pair decode stands at
  auto fetch_b = [&](int t) {  // t = the previous call's t or that + 1 (the end of the split re-fetches the last slab)
    if constexpr (TB == TB_PAIRSUM_RELU) {
      const bool adv = t != pf_t;
      pf_t = t;
      const int ni = pf_i + (adv ? BK : 0);
      const bool wrap = ni >= p.pairB;
      pf_i = wrap ? ni - p.pairB : ni;
      pf_j = wrap ? pf_j + 1 : pf_j;
      const float* src = p.B + (long)pf_i * p.ldb + n0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = bload4(src + (long)(8 * q) * p.ldb, b_lane);
      rb2 = bload4(p.B2 + (long)pf_j * p.ldb2 + n0, b2_lane);
    } else {
      const float* src = p.B + (r_begin + (long)t * BK) * p.ldb + n0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = bload4(src + (long)(8 * q) * p.ldb, b_lane);
    }
  };
  // make the fetched registers opaque at this point of the program: their consumers (and the s_waitcnt vmcnt that goes
  // with them) cannot be hoisted above it (gemm_engine.hpp pin4)
  auto pin_b = [&]() {
    asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb2));
  };
  // B image: inside a tile row the two 64-column groups of a wave's 128-column strip are interleaved pair-wise, so that
  // the four values a lane feeds to its four MFMA tiles (columns 2 l', 2 l' + 1 of group 0 and of group 1) are ONE 16-byte
  // read: column pair P = strip * 64 + g * 32 + l' sits at float strip * 128 + 4 l' + 2 g.  This thread holds the pairs
  // 2 lane and 2 lane + 1 (same strip and group, l' and l' + 1): two 8-byte writes 16 bytes apart.
  unsigned bw_addr = lds0 + 2u * TILEB + (unsigned)wave * ROWB +
                     (unsigned)((lane >> 5) * 128 + ((2 * lane) & 31) * 4 + ((lane & 31) >> 4) * 2) * 4u;
  asm volatile("" : "+v"(bw_addr));
  auto commit_b = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x2 lo, hi;
      if constexpr (TB == TB_AFFINE_RELU) {
        lo = pk_fma(rb[q].xy, bs.xy, bt.xy);
        hi = pk_fma(rb[q].zw, bs.zw, bt.zw);
      } else {
        lo = pk_add(rb[q].xy, rb2.xy);
        hi = pk_add(rb[q].zw, rb2.zw);
      }
      lds_write2(bw_addr + (BUF * TILEB + (unsigned)(8 * q) * ROWB), make_float2(relu_raw(lo.x), relu_raw(lo.y)));
      lds_write2(bw_addr + (BUF * TILEB + (unsigned)(8 * q) * ROWB + 16u), make_float2(relu_raw(hi.x), relu_raw(hi.y)));
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads are 8-byte: lane l takes columns (2 l', 2 l' + 1), l' = l % 32, of k-row 2 kk + l / 32 - MFMA tile 0 of
  // a wave owns the even and tile 1 the odd columns of its 64-wide strip (un-permuted in the epilogue)
  const int fcol = lane & 31;
  const int fk = lane >> 5;
  unsigned fa_addr = lds0 + (unsigned)fk * ROWB + (unsigned)(wm * 64 + 2 * fcol) * 4u;
  unsigned fb_addr = lds0 + 2u * TILEB + (unsigned)fk * ROWB + (unsigned)(wn * 128 + 4 * fcol) * 4u;
  asm volatile("" : "+v"(fa_addr), "+v"(fb_addr));

  // Fragment pipeline: (fa, fb) always hold the fragments of the NEXT k-pair to execute; the fragments of k-pair kk+1 are
  // read before the MFMAs of k-pair kk, so the LDS latency sits under the matrix pipe.  The loop is rotated across the
  // barrier like gemm_nt_dma_kernel's: the last k-pair's 8 MFMAs of a slab are issued AFTER the barrier, behind the first
  // fragment reads of the next slab.
  float2 fa;
  float4 fb;  // (group 0: x, y; group 1: z, w)
  auto read_pair = [&](auto buf_c, auto kk_c, float2& a, float4& b) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
    a = lds_read2(fa_addr + (BUF * TILEB + 2 * KK * ROWB));
    b = lds_read4(fb_addr + (BUF * TILEB + 2 * KK * ROWB));
  };
  auto mma8 = [&](float2 a, float4 b) {  // (same order of MFMAs per k-pair as gemm_tn_kernel: group 0's four tiles, then group 1's)
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
    acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.z, acc[0][2], 0, 0, 0);
    acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.w, acc[0][3], 0, 0, 0);
    acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.z, acc[1][2], 0, 0, 0);
    acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.w, acc[1][3], 0, 0, 0);
  };
  // k-pairs [KK0, KK1) of the slab in buffer BUF: entered with (fa, fb) = fragments of KK0, left with those of KK1
  // (KK1 < 16; the caller reads k-pair 0 of the next slab itself, after the barrier)
  auto compute = [&](auto buf_c, auto kk0_c, auto kk1_c) {
    constexpr int BUF = decltype(buf_c)::value, KK0 = decltype(kk0_c)::value, KK1 = decltype(kk1_c)::value;
    static_for<KK0, KK1>([&](auto kk_c) {
      constexpr int KK = decltype(kk_c)::value;
      float2 na;
      float4 nb;
      read_pair(std::integral_constant<int, BUF>{}, std::integral_constant<int, KK + 1>{}, na, nb);
      mma8(fa, fb);
      fa = na;
      fb = nb;
    });
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using K0 = integral_constant<int, 0>;
  using K8 = integral_constant<int, PN_TN_WAIT_KK>;  // k-pair in front of which the loads of slab t+1 are waited for
  using K15 = integral_constant<int, 15>;
  if (nsl > 0) {
    const int last = nsl - 1;
    issue_a(0, I0{});
    fetch_b(0);
    pin_b();
    commit_b(I0{});
    fetch_b(last < 1 ? last : 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_pair(I0{}, K0{}, fa, fb);
    // slab t < last out of buffer CUR: A(t+1) by DMA at the top (the idle buffer was last read in the previous slab,
    // barrier passed); late in the slab (before k-pair 14 of 16: measured best of 8 / 12 / 14 / 15) everything in flight is at least ~14k cycles old - this wave's share of
    // the A(t+1) DMA (invisible to hipcc: explicit wait) and the B(t+1) registers, fetched a whole slab ago; B(t+1) is
    // transformed and written under the second half of the MFMAs, B(t+2) goes out behind it and stays in flight across
    // the barrier.  Past the end the last slab is fetched again (branch-free; nobody reads it).
    // Keeping a task's 32 workgroups in step.  They share operand panels through the XCD's 4 MB L2, which holds ~10 slabs of
    // the task's stream; nothing couples them once every load is hidden (the better the prefetch, the freer they drift -
    // measured: fetch 0.49 TB per launch when they happen to stay together, 1.5-1.9 TB when not, same speed).  Every
    // PN_TN_SYNC_SLABS slabs workgroup thread 0 reports its arrival and waits until ALL 32 have reached the previous
    // checkpoint, so nobody is more than two checkpoints ahead.  The wait gives up after ~0.3 ms (it is an optimisation, never a dependency: if the 32
    // were not co-resident it must not hang), the other waves simply meet thread 0 at the slab's barrier.
    auto checkpoint = [&](int t) {
      // (scalar conditions first: the other seven waves skip this with one scalar branch; called where no load is in
      //  flight - right after the mid-slab wait - because hipcc answers a control-flow merge with conservative waits)
      if (sync_ctr != nullptr && (t & (PN_TN_SYNC_SLABS - 1)) == 0 && wave == 0 && lane == 0) {  // epoch e reports into slot e % 4
        const int e = t / PN_TN_SYNC_SLABS;
        __hip_atomic_fetch_add(sync_ctr + (e & 3), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e > 1) {
          const int* c = sync_ctr + ((e - 1) & 3);
          const int want = 32 * (((e - 1) >> 2) + 1);
          for (int spin = 0; spin < 1024 && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
            __builtin_amdgcn_s_sleep(8);
        }
      }
    };
    auto slab = [&](int t, auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value;
      using C = integral_constant<int, CUR>;
      using N = integral_constant<int, CUR ^ 1>;
      issue_a(t + 1, N{});
      __builtin_amdgcn_sched_barrier(0);
      compute(C{}, K0{}, K8{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pin_b();
      if constexpr (SYNC && CUR == 0) checkpoint(t);
      compute(C{}, K8{}, K15{});
      commit_b(N{});
      __builtin_amdgcn_sched_barrier(0);
      fetch_b(t + 2 < last ? t + 2 : last);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (k-pair 15's fragments are in registers: buffer CUR is free)
      __builtin_amdgcn_s_barrier();
      float2 na;
      float4 nb;
      read_pair(N{}, K0{}, na, nb);
      __builtin_amdgcn_sched_barrier(0);
      mma8(fa, fb);  // k-pair 15 of slab t, behind the first reads of slab t+1
      fa = na;
      fb = nb;
    };
    auto last_slab = [&](auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value;
      compute(integral_constant<int, CUR>{}, K0{}, K15{});
      mma8(fa, fb);
    };
    int t = 0;
    for (; t + 2 <= last; t += 2) {
      slab(t, I0{});
      slab(t + 1, I1{});
    }
#ifdef PN_TN_TAIL_RT
    // Experiment: the two compile-time copies of the last slab below make hipcc spill ~850-990 registers
    // in the TAIL (the 256-MFMA loop body is spill-free; the spills run once per workgroup and split).  Variant: the last slab
    // through a run-time buffer offset, as gemm_tn_bf16tr_kernel does - one copy of its code.
    if (t < last) slab(t, I0{});  // one more full slab: the last one then sits in buffer 1
    fa_addr += (unsigned)(last & 1) * TILEB;
    fb_addr += (unsigned)(last & 1) * TILEB;
    last_slab(I0{});
#else
    if (t < last) {  // one more full slab, then the last one sits in buffer 1
      slab(t, I0{});
      last_slab(I1{});
    } else {
      last_slab(I0{});
    }
#endif
  }

  float* out = p.Cpart + (long)split * p.M * p.ldc;
  const int hl = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = n0 + wn * 128 + g * 64 + 2 * fcol;  // columns n, n+1 <- tiles 2g, 2g+1
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * hl) + i;
        *reinterpret_cast<float2*>(out + (long)m * p.ldc + n) = make_float2(acc[i][2 * g][e], acc[i][2 * g + 1][e]);
      }
    }
  }
}

constexpr int TN_FAST_LDS_BYTES = 4 * 32768;

}  // namespace pn
