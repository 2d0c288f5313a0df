// A_k = W^T diag(vbar_k) W for all topics of a group - the ONE-WAVE-PER-SIMD form of gemm_tn_topics.h ("f16x3" arithmetic).
//
// What tn_topics_f16_kernel spends its time on (timing-only ablations, profiles/r03/README.md): 9.6 ms, of which the matrix pipe needs
// 3.9; without the per-topic split + LDS image writes 6.8 ms, without the MFMAs 6.5 ms, with one B fragment set per phase instead of
// one per topic 8.5 ms.  The scaled operand  y_k = vbar_kn w_nj  goes registers -> split -> LDS image -> transposing read -> MFMA once
// per topic, behind two barriers per 32-row chunk, with 160 accumulator registers per lane leaving room for two waves per SIMD only.
//
// Here the order is turned round: the raw f32 rows of W (the same for every topic) go to LDS once per chunk by DMA; a wave reads ITS
// 32 columns of them n-major - 8 consecutive n per lane, the B-fragment geometry of the 16x16x32 MFMA - into registers once per
// k-step, and per topic multiplies by the 8 row factors (an LDS broadcast read), splits in registers and feeds the MFMA directly:
// no per-topic LDS image, no transposing read of B, no phase barriers - one barrier per 64-row chunk.  A wave owns the B fragments of
// its columns alone (tile 64 (i) x 64 (j), 4 waves x 16 columns), so nothing is split twice inside a workgroup; the 64 x 16 x 10-topic
// accumulators are 160 registers per lane and live in the AGPR half of the file (launch bound 256 x 1: hipcc selects AGPR-form MFMAs
// only for kernels that may use more than 256 registers; with 64 x 32 columns per wave the 320 accumulators exceed the 256 AGPRs and
// it shuttles them through v_accvgpr moves - 1 500 per chunk).  With a single wave per SIMD the vector work hides inside the wave's own
// MFMA shadow: 2 VALU instructions per 16-cycle MFMA.
//
// LDS per buffer (two buffers, DMA of chunk c + 1 under the products of chunk c):
//   A   2 pieces x [32 rows][128 halfwords] image of gemm_tn_split_kernel (swizzle, transposing fragment read); the 128 "columns" are
//       the tile's 64 columns i for k-step 0 followed by the same 64 for k-step 1 (rows n + 32)
//   x   64 rows x 64 f32, as 16 blocks of four rows (one DMA instruction each) 1056 bytes apart: the 32 bytes of padding turn the
//       8-row stride between the four lane groups of a fragment read into a 16-bank offset - conflict-free ds_read_b32
//   v   [12][64 rows] row factors vbar_kn x block scale of the 10 topics (pre-scaled and zero-padded by tn1_scale_rows_kernel, so that
//       they too arrive by DMA: a register load that hipcc waits for with its own vmcnt would drain the DMA queue behind it)
// Three buffers: the DMA runs two chunks (~5 us) ahead - with one chunk of lookahead the HBM latency under load showed (timing-only
// ablation without the DMA: 11.4 -> 8.1 ms).
#pragma once
#include <utility>
#include "gemm_tn_topics.h"

namespace gdrf {

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}).  (A `#pragma unroll` of 20 stages x 24
// MFMAs with inline asm in the body is refused by hipcc's unroller - "loop not unrolled" - and the accumulator arrays land in scratch.)
template <int... Is, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int TN1_CH = 64;                        // rows per chunk
constexpr int TN1_XBLK = 1056;                    // bytes per four-row block of the x image
constexpr int TN1_A_BYTES = 2 * 32 * 128 * 2;     // 16 KB
constexpr int TN1_X_BYTES = 16 * TN1_XBLK;        // 16 896
constexpr int TN1_VROWS = 12;                     // table rows: 10 topics + 2 so that every wave issues the same three DMA requests
constexpr int TN1_V_BYTES = TN1_VROWS * TN1_CH * 4;  // 3 072
constexpr int TN1_BUF = TN1_A_BYTES + TN1_X_BYTES + TN1_V_BYTES;
constexpr int TN1_NBUF = 3;                       // chunks c + 1 and c + 2 are in flight under the products of chunk c
constexpr int TN1_DMA_PER_CHUNK = 11;             // DMA requests per wave and chunk: 4 (A) + 4 (x) + 3 (row factors)
constexpr int tn1_lds_bytes() { return TN1_NBUF * TN1_BUF; }

// 4-byte LDS-DMA (see glds16_asm): lane i's dword lands at lds_dst + 4 i
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void glds4_asm_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// vs[k][n] = vbar[k][n] x block scale of topic k for n < nrows, 0 for nrows <= n < lds (lds a multiple of 64): the row factors as the
// A_k kernel's DMA wants them - scaled, and zero past the last row so that a chunk may run over the end
__global__ void tn1_scale_rows_kernel(const float* __restrict__ vbar, int64_t ldk, int64_t nrows, float* __restrict__ vs, int64_t lds,
                                      const float* __restrict__ sc, int sidx_v) {
  const int k = blockIdx.y;
  const float scale = sc[sidx_v + 2 * k];
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < lds; n += (int64_t)gridDim.x * blockDim.x)
    vs[(int64_t)k * lds + n] = n < nrows ? vbar[(int64_t)k * ldk + n] * scale : 0.0f;
}

// tiles (I, J): 64-column blocks of the rows i and of the columns j, J <= I
__host__ __device__ inline int tn1_ntiles(int ncols) {
  const int nI = (ncols + 63) / 64;
  return nI * (nI + 1) / 2;
}

__global__ __launch_bounds__(256, 1) void tn_topics_w1_kernel(TNTopicsArgs g) {
  using E = _Float16;
  using V8 = f16x8;
  using V4 = f16x4;
  constexpr int KT = TNT_KT, PIECE = 32 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  // block -> (split, topic group, tile).  With a multiple of 8 splits every XCD (block mod 8) owns whole splits: the tiles of a split
  // run side by side on its CUs and share the rows they stream through that XCD's L2.
  const int kgroups = (g.K + KT - 1) / KT;
  const unsigned units = (unsigned)(g.ntiles * kgroups), bid = blockIdx.x;
  unsigned u; int sp;
  if ((g.nsplit & 7) == 0) { const unsigned idx = bid >> 3; sp = (int)(idx / units) * 8 + (int)(bid & 7u); u = idx % units; }
  else { sp = (int)(bid / units); u = bid % units; }
  const int tile = (int)(u % (unsigned)g.ntiles), gk = (int)(u / (unsigned)g.ntiles);
  int I = 0, J = tile;
  for (;; ++I) { if (J <= I) break; J -= I + 1; }
  const int i0 = I * 64, j0 = J * 64;
  const int k0 = gk * KT, kg = min(KT, g.K - k0);
  const int64_t r0 = (int64_t)sp * g.rows_per_split;
  int64_t r1 = r0 + g.rows_per_split; if (r1 > g.nrows) r1 = g.nrows;
  const int nch = r0 < r1 ? (int)((r1 - r0 + TN1_CH - 1) / TN1_CH) : 0;
  // this wave's 16 columns lie in the 32 x 32 sub-tile column 2 J + w / 2; its rows in the sub-tile rows 2 I (a = 0, 1) and 2 I + 1
  // (a = 2, 3); kept when on or below the diagonal (reduce_slabs_kernel mirrors the rest)
  const bool col_ok = (j0 + 16 * w) < g.ncols;
  const bool act0 = col_ok && (2 * I >= 2 * J + (w >> 1)), act1 = col_ok && (2 * I + 1 >= 2 * J + (w >> 1));

  f32x4 acc[KT][4];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[k][a] = f32x4{0, 0, 0, 0};

  const unsigned lds0 = lds_addr(smem);
  // ---- staging
  auto dma = [&](int c, int buf) {
    const unsigned base = lds0 + (unsigned)(buf * TN1_BUF);
    const int64_t n0 = r0 + (int64_t)c * TN1_CH;
    // A: this wave moves row blocks w and w + 4 (4 rows each) of both piece images
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rb = w + 4 * h, k = 4 * rb + (lane >> 4);
      const int c8 = (lane & 15) ^ (tnb_code(k) << 1);             // logical 8-column unit that lands at this lane's 16 bytes of the row
      int64_t n = n0 + 32 * (c8 >> 3) + k;
      n = n < g.nrows ? n : g.nrows - 1;
      const int col = (i0 + 8 * (c8 & 7) < g.ncols) ? i0 + 8 * (c8 & 7) : 0;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        glds16_asm(g.Ah + p * g.a_stride + n * g.lda + col, base + (unsigned)(p * PIECE * 2 + rb * 1024));
    }
    // x: four-row blocks 4 w .. 4 w + 3
    const int xcol = (j0 + 4 * (lane & 15) < g.ncols) ? j0 + 4 * (lane & 15) : 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int blk = 4 * w + t;
      int64_t n = n0 + 4 * blk + (lane >> 4);
      n = n < g.nrows ? n : g.nrows - 1;
      glds16_asm(g.B + n * g.ldb + xcol, base + (unsigned)(TN1_A_BYTES + blk * TN1_XBLK));
    }
    // row factors: table rows w, w + 4, w + 8 (rows 10 and 11 are filler: topic clamped)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int kk = w + 4 * t, ks = kk < kg ? kk : kg - 1;
      glds4_asm(g.vbar + (int64_t)(k0 + ks) * g.ldk + n0 + lane, base + (unsigned)(TN1_A_BYTES + TN1_X_BYTES + kk * TN1_CH * 4));
    }
  };
  // A fragments (geometry of gemm_tn_split_kernel): two transposing reads of 4 rows x 16 columns each
  const int fq = lr >> 2, fp = lr & 3, fcode = ((lg & 1) << 2) | fq;
  const int fbase = (8 * lg + fq) * 32 + fp;
  auto frag = [&](const E* img, int g16) -> V8 {
    const int seg = fbase + ((g16 ^ fcode) << 2);
    const V4 lo = tr_read(img + seg * 4);
    const V4 hi = tr_read(img + (seg + 128) * 4);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // two neighbouring rows (xa va, xb vb) of one column -> packed fp16 pairs H = (h_a, h_b), L = (l'_a, l'_b), l' = 2^11 (x v - h):
  // see split2 of tn_topics_f16_kernel (there the two elements share the row factor; here they are consecutive rows)
  const float c2048 = 2048.0f;
  auto split2v = [&](float xa, float xb, float va, float vb, unsigned& H, unsigned& L) {
    float ra, rb2;
    asm("v_fma_mixlo_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %6, 0\n\t"
        "v_fma_mix_f32 %1, %3, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %2, %5, %6, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(H), "=&v"(ra), "=&v"(rb2) : "v"(xa), "v"(va), "v"(xb), "v"(vb));
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0\n\t"
        "s_nop 1"
        : "=&v"(L) : "v"(ra), "v"(rb2), "v"(c2048));
  };

  // the same for two pairs at once, the two dependent chains interleaved
  auto split4v = [&](float xa, float xb, float xc, float xd, float va, float vb, float vc, float vd, unsigned& H0, unsigned& L0, unsigned& H1, unsigned& L1) {
    float r0, r1, r2, r3;
    asm("v_fma_mixlo_f16 %0, %8, %9, 0\n\t"
        "v_fma_mixlo_f16 %2, %12, %13, 0\n\t"
        "v_fma_mixhi_f16 %0, %10, %11, 0\n\t"
        "v_fma_mixhi_f16 %2, %14, %15, 0\n\t"
        "v_fma_mix_f32 %4, %8, %9, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %6, %12, %13, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %10, %11, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %7, %14, %15, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, %16, 0\n\t"
        "v_fma_mixlo_f16 %3, %6, %16, 0\n\t"
        "v_fma_mixhi_f16 %1, %5, %16, 0\n\t"
        "v_fma_mixhi_f16 %3, %7, %16, 0\n\t"
        "s_nop 1"
        : "=&v"(H0), "=&v"(L0), "=&v"(H1), "=&v"(L1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(xa), "v"(va), "v"(xb), "v"(vb), "v"(xc), "v"(vc), "v"(xd), "v"(vd), "v"(c2048));
  };

  // wait until the DMA of the next chunk has landed: everything but the requests of the chunk after it (issued last) is complete
  auto wait_next = [&](bool two_ahead) {
    if (two_ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(TN1_DMA_PER_CHUNK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };
  if (nch > 0) {
    dma(0, 0);
    if (nch > 1) dma(1, 1);
    wait_next(nch > 1);
    gdrf_raw_barrier();
    // waves whose sub-tiles all lie above the diagonal only stage.  (A separate loop, not a branch around the products: with the branch
    // hipcc carries the accumulators in VGPRs and copies each one into an AGPR in front of its MFMAs and back - 32 moves per 12 MFMAs.)
    if (!act1) {
      int bn = 2;                                // buffer of chunk c + 2
      for (int c = 0; c < nch; ++c) {
        const bool two = c + 2 < nch;
        if (two) dma(c + 2, bn);
        bn = bn == 2 ? 0 : bn + 1;
        wait_next(two);
        gdrf_raw_barrier();
      }
      return;
    }
    int buf = 0, bn = 2;
    for (int c = 0; c < nch; ++c) {
      const bool two = c + 2 < nch;
#if defined(GDRF_DIAG) && (GDRF_W1_ABLATE == 3 || GDRF_W1_ABLATE == 4)        // timing-only: no DMA after the first chunks (wrong results)
      if (two && c < 1) dma(c + 2, bn);
#else
      if (two) dma(c + 2, bn);
#endif
      {
        const E* As = reinterpret_cast<const E*>(smem + buf * TN1_BUF);
        const float* xs = reinterpret_cast<const float*>(smem + buf * TN1_BUF + TN1_A_BYTES);
        const float* vt = reinterpret_cast<const float*>(smem + buf * TN1_BUF + TN1_A_BYTES + TN1_X_BYTES);
        // every LDS operand of the chunk that does not depend on the topic is read up front: A fragments and x columns of both k-steps
        V8 fa[2][4][2];
        float xv[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p) fa[s][a][p] = frag(As + p * PIECE, 4 * s + a);
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[s][e] = xs[(8 * s + 2 * lg + (e >> 2)) * (TN1_XBLK / 4) + (e & 3) * 64 + 16 * w + lr];
        }
        // stage t = (k-step s = t / KT, topic k = t % KT), software-pipelined by hand - one wave per SIMD has nobody to hide its waits:
        //   row factors of stage t + 2 (LDS broadcast read)  |  split of stage t + 1 (VALU)  |  the 12 MFMAs of stage t
        // with a scheduling barrier between stages (without it hipcc hoists the loads and splits of ALL stages to the top and spills)
        constexpr int NST = 2 * KT;
        f32x4 vv[3][2];
        auto load_vv = [&](int t) {
          const int s = t / KT, k = t % KT;
          vv[t % 3][0] = *reinterpret_cast<const f32x4*>(vt + k * TN1_CH + 32 * s + 8 * lg);
          vv[t % 3][1] = *reinterpret_cast<const f32x4*>(vt + k * TN1_CH + 32 * s + 8 * lg + 4);
        };
        V8 fbh[2], fbl[2];
        auto split_b = [&](int t) {
          const int s = t / KT;
          const f32x4 v0 = vv[t % 3][0], v1 = vv[t % 3][1];
          const float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          unsigned H[4], L[4];
#ifdef GDRF_W1_SPLIT4
#pragma unroll
          for (int h2 = 0; h2 < 4; h2 += 2) split4v(xv[s][2 * h2], xv[s][2 * h2 + 1], xv[s][2 * h2 + 2], xv[s][2 * h2 + 3], f[2 * h2], f[2 * h2 + 1],
                                                    f[2 * h2 + 2], f[2 * h2 + 3], H[h2], L[h2], H[h2 + 1], L[h2 + 1]);
#else
#pragma unroll
          for (int h2 = 0; h2 < 4; ++h2) split2v(xv[s][2 * h2], xv[s][2 * h2 + 1], f[2 * h2], f[2 * h2 + 1], H[h2], L[h2]);
#endif
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          fbh[t & 1] = __builtin_bit_cast(V8, u32x4{H[0], H[1], H[2], H[3]});
          fbl[t & 1] = __builtin_bit_cast(V8, u32x4{L[0], L[1], L[2], L[3]});
        };
        load_vv(0);
        load_vv(1);
        split_b(0);
        V8 fa2[4];                             // 2^-11 x the high piece: partner of the B operand's up-scaled low piece
#pragma unroll
        for (int t = 0; t < NST; ++t) {
          const int s = t / KT, k = t % KT;
          __builtin_amdgcn_sched_barrier(0);
          if (k == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) fa2[a] = fa[s][a][0] * (E)0.00048828125f;
          }
#if defined(GDRF_DIAG) && (GDRF_W1_ABLATE == 1 || GDRF_W1_ABLATE == 4)      // timing-only: no row factors, no splits (wrong results)
          if (c == 0) {
#endif
          if (t + 2 < NST) load_vv(t + 2);
          if (t + 1 < NST) split_b(t + 1);
#if defined(GDRF_DIAG) && (GDRF_W1_ABLATE == 1 || GDRF_W1_ABLATE == 4)
          }
#endif
#if defined(GDRF_DIAG) && GDRF_W1_ABLATE == 2      // timing-only: no MFMAs (wrong results)
          asm volatile("" :: "v"(fbl[t & 1]), "v"(fbh[t & 1]), "v"(fa2[0]));
          continue;
#endif
          // product order: (h_a, l_b), (l_a, h_b), (h_a, h_b), each over the four accumulators of the topic
#pragma unroll
          for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int a = 0; a < 4; ++a)
              acc[k][a] = SplitF16::mma(x == 0 ? fa2[a] : fa[s][a][x == 1 ? 1 : 0], x == 0 ? fbl[t & 1] : fbh[t & 1], acc[k][a]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_next(two);                      // this wave's DMA of chunk c + 1 has landed
      gdrf_raw_barrier();
      buf = buf == 2 ? 0 : buf + 1;
      bn = bn == 2 ? 0 : bn + 1;
    }
  }
  if (!act1) return;                            // (no rows at all)
  const float una = g.sc[g.sidx_a + 1];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < kg) {
      const float un = una * g.sc[g.sidx_v + 2 * (k0 + k) + 1];
      float* out = g.slab + ((int64_t)sp * g.K + k0 + k) * (int64_t)g.ncols * g.ncols;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a < 2 && !act0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + 16 * a + 4 * lg + r;
          const int j = j0 + 16 * w + lr;
          if (i < g.ncols && j < g.ncols) out[(int64_t)i * g.ncols + j] = acc[k][a][r] * un;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// The same contraction with 128 rows i per wave: tile 128 (i) x 64 (j), 4 waves x 16 columns.
//
// Why: one SIMD has ONE vector issue port, and a 16x16x32 MFMA holds it for 8 of its 16 cycles (MI355X_MICROARCH.md, row 'vector-
// instruction ISSUE cost') - per MFMA there are 8 cycles of VALU issue to hide in, whether one wave or two share the SIMD.  The split of
// the row-scaled operand costs the same per B fragment whatever the number of A tiles it is multiplied with, so the VALU cycles per
// MFMA go with 1 / (rows i per wave): tn_topics_w1_kernel (64 rows, 24 v_fma_mix at 8 cycles per 12 MFMAs) has 16 cycles of split per
// MFMA and runs VALU-bound (8.85 ms; 5.0 ms with the splits ablated, 5.0 ms with the MFMAs ablated).  Here a B fragment meets 8 A
// tiles, and the split is rewritten on 4-cycle instructions (v_mul, v_and, v_sub, v_cvt_pk - 10 per element pair instead of 6
// v_fma_mix at 8): 160 cycles of split + 192 of MFMA issue hold per 384-cycle stage.
//
// Accumulators: 8 tiles x 10 topics x 4 = 320 registers per lane.  hipcc puts every MFMA of a 512-register kernel into the AGPR form
// and then has 256 AGPRs for 320 accumulators (it shuttles them: 1 500 v_accvgpr moves per chunk), so the MFMAs are issued from inline
// asm with the register file chosen per topic: topics 0 - 7 accumulate in AGPRs, topics 8 and 9 in VGPRs.  Inline asm is outside
// hipcc's hazard tables; the three cases that matter are covered by construction: an accumulator is touched again 8 MFMAs later (24
// products of a stage in product-major order), the B fragments of stage t + 1 are written at least 6 MFMAs before stage t + 1 starts,
// and the epilogue waits two s_nop 15 before it reads the accumulators.
//
// The split:  y = x v;  h = y with the mantissa cut to 11 bits (exactly what fp16 holds: v_and, then an exact v_cvt_pkrtz);  l' = 2^11
// (y - h), the difference exact in f32, rounded to nearest by v_cvt_pk_f16_f32.  |l'| < 2 |h| 2^-10 2^11: same range and the same 22
// bits as the RNE split of tn_topics_f16_kernel (the cut h is one ulp coarser, the residual carries it).
constexpr int TN2_A_BYTES = 2 * 2 * 32 * 128 * 2;            // [k-step][piece][32][128] halfwords: 32 KB
constexpr int TN2_BUF = TN2_A_BYTES + TN1_X_BYTES + TN1_V_BYTES;
constexpr int TN2_DMA_PER_CHUNK = 15;                          // 8 (A) + 4 (x) + 3 (row factors)
constexpr int tn2_lds_bytes() { return TN1_NBUF * TN2_BUF; }
#ifdef GDRF_TN2_STAMPS   // diagnostic builds only: s_memtime stamps of one workgroup's wave 0 (tools/tn2_stamps.py)
__device__ unsigned long long* g_tn2_stamps = nullptr;       // [chunk][8]
#define TN2_STAMP(slot) do { if (g_tn2_stamps && blockIdx.x == 1000 && threadIdx.x == 0 && c < 256) g_tn2_stamps[c * 8 + (slot)] = split_stamp(); } while (0)
#else
#define TN2_STAMP(slot) do {} while (0)
#endif

// AMIN: the first A tile (16 rows) the workgroups of this launch multiply.  The tiles J = 2 I + 1 start 64 columns right of their first
// row, so their rows i0 .. i0 + 63 (A tiles 0 - 3) lie entirely above the diagonal: 12 MFMAs per stage instead of 24 (such a stage is as long
// as its split - 20 pieces, 160 issue cycles + 96 of MFMA hold - not half as long).  They run as a launch of their own (AMIN = 4: g.ntiles
// of them, tile t = (I = t, J = 2 t + 1)) behind the launch of all the others (AMIN = 0): a branch around the asm MFMAs would make hipcc
// copy accumulators, and a lambda instantiated twice inside one kernel left its captures in scratch memory.
__host__ __device__ inline int tn2_ntiles_half(int ncols) {            // tiles (I, 2 I + 1) inside the matrix
  const int nI = (ncols + 127) / 128, nJ = (ncols + 63) / 64;
  int t = 0;
  for (int I = 0; I < nI; ++I) t += (2 * I + 1 < nJ) ? 1 : 0;
  return t;
}
template <int AMIN>
__global__ __launch_bounds__(256, 1) void tn_topics_w2_kernel(TNTopicsArgs g) {
  using E = _Float16;
  using V8 = f16x8;
  using V4 = f16x4;
  constexpr int KT = TNT_KT, PIECE = 32 * 128, KA = 8;       // KA: topics whose accumulators live in AGPRs
  static_assert(KT == 10, "register plan below is for 10 topics");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int kgroups = (g.K + KT - 1) / KT;
  const unsigned units = (unsigned)(g.ntiles * kgroups), bid = blockIdx.x;
  unsigned u; int sp;
  if ((g.nsplit & 7) == 0) { const unsigned idx = bid >> 3; sp = (int)(idx / units) * 8 + (int)(bid & 7u); u = idx % units; }
  else { sp = (int)(bid / units); u = bid % units; }
  const int tile = (int)(u % (unsigned)g.ntiles), gk = (int)(u / (unsigned)g.ntiles);
  int I = 0, J = tile;
  if (AMIN == 4) { I = tile; J = 2 * tile + 1; }
  else {
    // the tiles (I, J), J <= 2 I + 1, J < nJ, without those of the other launch (J == 2 I + 1)
    const int nJ = (g.ncols + 63) / 64;
    for (;; ++I) { const int cnt = 2 * I + 1 < nJ ? 2 * I + 1 : nJ; if (J < cnt) break; J -= cnt; }
  }
  const int i0 = I * 128, j0 = J * 64;
  const int k0 = gk * KT, kg = min(KT, g.K - k0);
  const int64_t r0 = (int64_t)sp * g.rows_per_split;
  int64_t r1 = r0 + g.rows_per_split; if (r1 > g.nrows) r1 = g.nrows;
  const int nch = r0 < r1 ? (int)((r1 - r0 + TN1_CH - 1) / TN1_CH) : 0;
  const bool col_ok = (j0 + 16 * w) < g.ncols;
  const int bj = 2 * J + (w >> 1);                             // 32 x 32 sub-tile column of this wave; its rows: 4 I + a / 2

  f32x4 accA[KA][8], accV[KT - KA][8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
#pragma unroll
    for (int k = 0; k < KA; ++k) accA[k][a] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < KT - KA; ++k) accV[k][a] = f32x4{0, 0, 0, 0};
  }

  const unsigned lds0 = lds_addr(smem);
  auto dma_edge = [&](int c, int buf) {                 // chunks that run past the last row: rows clamped per lane
    const unsigned base = lds0 + (unsigned)(buf * TN2_BUF);
    const int64_t n0 = r0 + (int64_t)c * TN1_CH;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rb = w + 4 * h, k = 4 * rb + (lane >> 4);
        const int c8 = (lane & 15) ^ (tnb_code(k) << 1);
        int64_t n = n0 + 32 * s + k;
        n = n < g.nrows ? n : g.nrows - 1;
        const int col = (i0 + 8 * c8 < g.ncols) ? i0 + 8 * c8 : 0;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          glds16_asm(g.Ah + p * g.a_stride + n * g.lda + col, base + (unsigned)((s * 2 + p) * PIECE * 2 + rb * 1024));
      }
    const int xcol = (j0 + 4 * (lane & 15) < g.ncols) ? j0 + 4 * (lane & 15) : 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int blk = 4 * w + t;
      int64_t n = n0 + 4 * blk + (lane >> 4);
      n = n < g.nrows ? n : g.nrows - 1;
      glds16_asm(g.B + n * g.ldb + xcol, base + (unsigned)(TN2_A_BYTES + blk * TN1_XBLK));
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int kk = w + 4 * t, ks = kk < kg ? kk : kg - 1;
      glds4_asm(g.vbar + (int64_t)(k0 + ks) * g.ldk + n0 + lane, base + (unsigned)(TN2_A_BYTES + TN1_X_BYTES + kk * TN1_CH * 4));
    }
  };
  // chunks inside the rows: wave-uniform 64-bit bases (scalar arithmetic) + one constant 32-bit byte offset per lane and operand kind.
  // (The per-lane 64-bit address arithmetic and clamps of dma_edge are ~100 vector instructions per chunk, issued with the matrix pipe idle.)
  unsigned voff_a, voff_x;
  {
    const int k = 4 * w + (lane >> 4);
    const int c8 = (lane & 15) ^ (tnb_code(k) << 1);          // rows k and k + 16 share the swizzle code
    const int col = (i0 + 8 * c8 < g.ncols) ? i0 + 8 * c8 : 0;
    voff_a = (unsigned)(((int64_t)(lane >> 4) * g.lda + col) * 2);
    const int xcol = (j0 + 4 * (lane & 15) < g.ncols) ? j0 + 4 * (lane & 15) : 0;
    voff_x = (unsigned)(((int64_t)(lane >> 4) * g.ldb + xcol) * 4);
  }
  const unsigned voff_v = (unsigned)lane * 4u;
  auto dma_is_edge = [&](int c) { return r0 + (int64_t)c * TN1_CH + TN1_CH > g.nrows; };
  // request q = 0 .. 14 of chunk c (not an edge chunk): 8 of A, 4 of x, 3 of the row factors
  auto dma_piece = [&](int c, int buf, int q) {
    const int64_t n0 = r0 + (int64_t)c * TN1_CH;
    const unsigned base = lds0 + (unsigned)(buf * TN2_BUF);
    if (q < 8) {
      const int s = q >> 2, h = (q >> 1) & 1, p = q & 1;
      glds16_asm_s(g.Ah + p * g.a_stride + (n0 + 32 * s + 16 * h + 4 * w) * g.lda, voff_a, base + (unsigned)((s * 2 + p) * PIECE * 2 + (w + 4 * h) * 1024));
    } else if (q < 12) {
      const int t = q - 8;
      glds16_asm_s(g.B + (n0 + 4 * (4 * w + t)) * g.ldb, voff_x, base + (unsigned)(TN2_A_BYTES + (4 * w + t) * TN1_XBLK));
    } else {
      const int kk = w + 4 * (q - 12), ks = kk < kg ? kk : kg - 1;
      glds4_asm_s(g.vbar + (int64_t)(k0 + ks) * g.ldk + n0, voff_v, base + (unsigned)(TN2_A_BYTES + TN1_X_BYTES + kk * TN1_CH * 4));
    }
  };
  auto dma = [&](int c, int buf) {
    if (dma_is_edge(c)) { dma_edge(c, buf); return; }
#pragma unroll
    for (int q = 0; q < TN2_DMA_PER_CHUNK; ++q) dma_piece(c, buf, q);
  };
  const int fq = lr >> 2, fp = lr & 3, fcode = ((lg & 1) << 2) | fq;
  const int fbase = (8 * lg + fq) * 32 + fp;
  auto frag = [&](const E* img, int g16) -> V8 {
    const int seg = fbase + ((g16 ^ fcode) << 2);
    const V4 lo = tr_read(img + seg * 4);
    const V4 hi = tr_read(img + (seg + 128) * 4);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // (xa va, xb vb): two consecutive rows of one column -> packed fp16 (h_a, h_b) and (l'_a, l'_b); see the header
  const unsigned kmask = 0xffffe000u;               // constants in scalar registers: a 32-bit literal makes the instruction two dwords
  const float k2048 = 2048.0f;
  // the split in five two-instruction pieces, so that it can be dealt out between single MFMAs: a wave issues in order and the matrix pipe
  // queues next to nothing - behind six back-to-back MFMAs the ten instructions of a whole split ran with the pipe idle (stamped: 587
  // cycles per 24-MFMA stage instead of 384)
  auto split_piece = [&](int q, float xa, float xb, float va, float vb, float& y0, float& y1, float& h0, float& h1, unsigned& H, unsigned& L) {
    if (q == 0) asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(y0), "=&v"(y1) : "v"(xa), "v"(va), "v"(xb), "v"(vb));
    if (q == 1) asm volatile("v_and_b32 %0, %2, %3\n\tv_and_b32 %1, %2, %4" : "=&v"(h0), "=&v"(h1) : "s"(kmask), "v"(y0), "v"(y1));
    if (q == 2) asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %3" : "+v"(y0), "+v"(y1) : "v"(h0), "v"(h1));
    if (q == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %1" : "=&v"(H), "+v"(y0) : "v"(h0), "v"(h1), "s"(k2048));
    if (q == 4) asm volatile("v_mul_f32 %1, %3, %1\n\tv_cvt_pk_f16_f32 %0, %2, %1" : "=&v"(L), "+v"(y1) : "v"(y0), "s"(k2048));
  };
  auto mfma_a = [&](f32x4& acc, const V8& a, const V8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); };
  auto mfma_v = [&](f32x4& acc, const V8& a, const V8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); };
  auto wait_next = [&](bool two_ahead) {
    if (two_ahead) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(TN2_DMA_PER_CHUNK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };

  if (nch > 0) {
    constexpr int NA = 8 - AMIN, NM = 3 * NA, NPS = NM - 2;     // NPS: MFMA slots that carry split pieces
    dma(0, 0);
    if (nch > 1) dma(1, 1);
    wait_next(nch > 1);
    gdrf_raw_barrier();
    int buf = 0, bn = 2;
    for (int c = 0; c < nch; ++c) {
      const bool two = c + 2 < nch;
      TN2_STAMP(0);
      // the DMA of chunk c + 2: one request per stage below, in the stage's last MFMAs (an edge chunk - rows clamped per lane - at once)
      const bool spread = two && !dma_is_edge(c + 2);
      if (two && !spread) dma_edge(c + 2, bn);
      TN2_STAMP(1);
      const E* As = reinterpret_cast<const E*>(smem + buf * TN2_BUF);
      const float* xs = reinterpret_cast<const float*>(smem + buf * TN2_BUF + TN2_A_BYTES);
      const float* vt = reinterpret_cast<const float*>(smem + buf * TN2_BUF + TN2_A_BYTES + TN1_X_BYTES);
      // behind the barrier every LDS read of the chunk's first stage is issued before anything waits for one: the A fragments of k-step 0
      // first (the pieces of the first split used to run in front of them: their latency showed once per chunk)
      V8 fah[8], fal[8], fa2[8];
#pragma unroll
      for (int a = AMIN; a < 8; ++a) { fah[a] = frag(As, a); fal[a] = frag(As + PIECE, a); }
      float xv[2][8];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[s][e] = xs[(8 * s + 2 * lg + (e >> 2)) * (TN1_XBLK / 4) + (e & 3) * 64 + 16 * w + lr];
      constexpr int NST = 2 * KT;
      f32x4 vv[3][2];
      auto load_vv = [&](int t) {
        const int s = t / KT, k = t % KT;
        vv[t % 3][0] = *reinterpret_cast<const f32x4*>(vt + k * TN1_CH + 32 * s + 8 * lg);
        vv[t % 3][1] = *reinterpret_cast<const f32x4*>(vt + k * TN1_CH + 32 * s + 8 * lg + 4);
      };
      unsigned Hh[2][4], Ll[2][4];                 // packed B fragments of the stage being multiplied and of the next one
      float y0, y1, h0, h1;
      auto piece = [&](int t, int q20) {           // piece q20 = 0 .. 19 of stage t's split: element pair q20 / 5, step q20 % 5
        const int s = t / KT, g2 = q20 / 5;
        const f32x4 v = vv[t % 3][g2 >> 1];
        split_piece(q20 % 5, xv[s][2 * g2], xv[s][2 * g2 + 1], v[2 * (g2 & 1)], v[2 * (g2 & 1) + 1], y0, y1, h0, h1, Hh[t & 1][g2], Ll[t & 1][g2]);
      };
      load_vv(0);
      load_vv(1);
#pragma unroll
      for (int a = AMIN; a < 8; ++a) fa2[a] = fah[a] * (E)0.00048828125f;
#pragma unroll
      for (int q = 0; q < 20; ++q) piece(0, q);
      // A VGPR written by a VALU instruction must not be read by an MFMA within the next two wait states; hipcc pads that for its own MFMAs,
      // not for inline asm.  Every operand that compiler-generated VALU code may have touched (the packed fragment tuples, the 2^-11 copies)
      // therefore passes through an asm statement that pins the point where it exists: `s_nop 1` behind it, or two MFMAs of the stage before.
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      V8 fbh = __builtin_bit_cast(V8, u32x4{Hh[0][0], Hh[0][1], Hh[0][2], Hh[0][3]});
      V8 fbl = __builtin_bit_cast(V8, u32x4{Ll[0][0], Ll[0][1], Ll[0][2], Ll[0][3]});
      asm volatile("s_nop 1" : "+v"(fbh), "+v"(fbl));
      static_for<NST>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int s = t / KT, k = t % KT;
        if (k == 0) TN2_STAMP(2 + 2 * s);
        if (k == 1) TN2_STAMP(3 + 2 * s);
        if (k == 0) {
          if (s == 1) {
#pragma unroll
            for (int a = AMIN; a < 8; ++a) { fah[a] = frag(As + 2 * PIECE, a); fal[a] = frag(As + 3 * PIECE, a); }
#pragma unroll
            for (int a = AMIN; a < 8; ++a) fa2[a] = fah[a] * (E)0.00048828125f;
          }
          if (AMIN == 0) asm volatile("s_nop 1" : "+v"(fa2[0]), "+v"(fa2[1]), "+v"(fa2[2]), "+v"(fa2[3]), "+v"(fa2[4]), "+v"(fa2[5]), "+v"(fa2[6]), "+v"(fa2[7]));
          else asm volatile("s_nop 1" : "+v"(fa2[4]), "+v"(fa2[5]), "+v"(fa2[6]), "+v"(fa2[7]));
        }
        if (t + 2 < NST) load_vv(t + 2);
        const V8 fbh_t = fbh, fbl_t = fbl;
        // NM = 24 (12) products, product-major: (h_a, l_b) x NA tiles, (l_a, h_b) x NA, (h_a, h_b) x NA; the 20 pieces of the next stage's split
        // behind the first NM - 2 of them (one piece = 2 VALU instructions: 8 issue cycles + the MFMA's own 8 = its 16 pipe cycles)
        static_for<NM>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          constexpr int x = m / NA, a = AMIN + m % NA;
          const V8& fa = x == 0 ? fa2[a] : (x == 1 ? fal[a] : fah[a]);
          const V8& fb = x == 0 ? fbl_t : fbh_t;
          if (k < KA) mfma_a(accA[k < KA ? k : 0][a], fa, fb); else mfma_v(accV[k >= KA ? k - KA : 0][a], fa, fb);
          if (m < NPS && t + 1 < NST) {
#pragma unroll
            for (int q = m * 20 / NPS; q < (m + 1) * 20 / NPS; ++q) piece(t + 1, q);
          }
          if (m == NPS - 1 && t + 1 < NST) {          // the next stage's fragment tuples exist from here on: two MFMAs follow
            fbh = __builtin_bit_cast(V8, u32x4{Hh[(t + 1) & 1][0], Hh[(t + 1) & 1][1], Hh[(t + 1) & 1][2], Hh[(t + 1) & 1][3]});
            fbl = __builtin_bit_cast(V8, u32x4{Ll[(t + 1) & 1][0], Ll[(t + 1) & 1][1], Ll[(t + 1) & 1][2], Ll[(t + 1) & 1][3]});
            asm volatile("" : "+v"(fbh), "+v"(fbl));
          }
          if (m == NM - 2 && spread && t < TN2_DMA_PER_CHUNK) dma_piece(c + 2, bn, t);
        });
      });
      TN2_STAMP(6);
      wait_next(two);
      gdrf_raw_barrier();
      TN2_STAMP(7);
      buf = buf == 2 ? 0 : buf + 1;
      bn = bn == 2 ? 0 : bn + 1;
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results (inline asm: outside hipcc's hazard tables)
  if (!col_ok) return;
  const float una = g.sc[g.sidx_a + 1];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < kg) {
      const float un = una * g.sc[g.sidx_v + 2 * (k0 + k) + 1];
      float* out = g.slab + ((int64_t)sp * g.K + k0 + k) * (int64_t)g.ncols * g.ncols;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        if (4 * I + (a >> 1) < bj) continue;                   // sub-tile above the diagonal: reduce_slabs_kernel mirrors it from below
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + 16 * a + 4 * lg + r;
          const int j = j0 + 16 * w + lr;
          const float v = k < KA ? accA[k < KA ? k : 0][a][r] : accV[k >= KA ? k - KA : 0][a][r];
          if (i < g.ncols && j < g.ncols) out[(int64_t)i * g.ncols + j] = v * un;
        }
      }
    }
  }
}

}  // namespace gdrf
