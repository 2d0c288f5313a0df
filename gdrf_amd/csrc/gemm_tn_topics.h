// A_k = W^T diag(vbar_k) W for ALL topics of a group in one pass over the observations ("f16x3" arithmetic, gemm_split.h).
//
// gemm_tn_split_kernel gives every (128 x 128 tile, topic, row split) its own workgroup, so each of them streams its rows of
// W - 1 KB per row - and the launch moves 100 x N x 1 KB = 100 GB at the headline size for 4 GB of distinct data (measured:
// FETCH_SIZE 85-88 GB, L2 hit rate 17 %, 7 TB/s: the kernel runs at the memory system's rate, not the matrix pipe's).  The
// reduction index n is shared by all topics and only the per-row factor vbar_kn differs, so here ONE workgroup owns a
// 128 (i) x 64 (j) output tile for up to KT = 10 topics - 320 KB of f32 accumulators, 160 registers per lane at 512 threads,
// which is what a CU's register file can hold - and streams W once for all of them: 20 tiles x N x 768 B = 15 GB.
//
// Per 32-row chunk: the A operand (the fp16 pieces of W, columns i) arrives by LDS-DMA (issued from asm, so that hipcc does
// not drain it in front of the next LDS read), double-buffered; the f32 rows of W for the j columns are loaded ONCE (one
// f32x4 per lane) and then scaled by vbar_kn x block scale and split into fp16 pieces per topic.  Topics go two at a time:
// the 64 + 64 columns of a topic pair form one 32 x 128 piece image (the image / swizzle / transposing-read geometry of
// gemm_tn_split_kernel), double-buffered, ONE barrier per pair: while the matrix cores multiply pair p, the vector units
// split pair p + 1.  The A fragments of a chunk stay in registers for all pairs.
// Waves: 8 = 4 (i) x 2 (j), each a 32 x 32 sub-tile per topic.  Symmetry at 32-column granularity: sub-tiles above the
// diagonal are skipped (their waves only stage), reduce_slabs_kernel mirrors them from below.
//
// Range of the scaled operand y = vbar_kn w_nj x block scale.  Its block scale comes from max_n |vbar_kn|, an outlier statistic: the rows
// that carry the sum sit 2^10 and more below it (measured on a contracted posterior: typical |y| = 2^-8 .. 2^-2 for a maximum of 2^13), and
// with y = h + l in plain fp16 pieces the low piece l ~ 2^-11 y drops into fp16's subnormals below |y| = 2^-3: the contraction then kept
// ~15 bits where the other arithmetic modes keep 22 (A_k error 5.9e-5 against 4e-7).  So the low piece is stored as l' = 2^11 l - the
// magnitude of y itself, a normal fp16 down to |y| = 2^-14 - and the product that contains it, h_a l_b, is issued as (2^-11 h_a) l'_b with
// the A fragment scaled in registers once per chunk (exact unless h_a itself is below 2^-3, i.e. |w| < 1e-4 sqrt(variance): those
// terms' corrections are second order).  Full 22 bits over 29 binades (8.7 decades) instead of 18 (5.4).
#pragma once
#include "gemm_split.h"

namespace gdrf {

constexpr int TNT_KT = 10;                    // topics per workgroup (accumulator budget)

struct TNTopicsArgs {
  const _Float16* Ah; int64_t a_stride; int64_t lda;   // fp16 pieces of W: [p][n][lda]
  const float* B; int64_t ldb;                          // f32 W: [n][ldb]
  const float* vbar; int64_t ldk;                       // [K][ldk]
  int64_t nrows, rows_per_split;                        // rows_per_split multiple of 32
  int ncols;                                            // Mp
  float* slab;                                          // [nsplit][K][ncols][ncols]
  int K, nsplit, ntiles;
  const float* sc; int sidx_a, sidx_v;                  // block scales (SplitLay): W pieces, vbar_k w operand (pair stride 2)
};

// tiles (I, J): 128-column block I of the rows i, 64-column block J <= 2 I + 1 of the columns j
__host__ __device__ inline int tnt_ntiles(int ncols) {
  const int nI = (ncols + 127) / 128, nJ = (ncols + 63) / 64;
  int t = 0;
  for (int I = 0; I < nI; ++I) t += (2 * I + 2 < nJ ? 2 * I + 2 : nJ);
  return t;
}

// PPH = topic pairs per phase (one barrier per phase).  With one pair a wave issues 24 MFMAs between barriers (384 pipe cycles against
// ~300 of barrier and ramp); with three the ten topics of a chunk are two phases (3 + 2 pairs) instead of five, at 2 x PPH B images.
constexpr int tnt_lds_bytes(int pph) { return 2 * 2 * 32 * 128 * 2 + 2 * pph * 2 * 32 * 128 * 2 + 2 * TNT_KT * 32 * 4; }
template <int PPH>
__global__ __launch_bounds__(512, 2) void tn_topics_f16_kernel(TNTopicsArgs g) {
  using SP = SplitF16;
  using E = _Float16;
  using V8 = f16x8;
  using V4 = f16x4;
  constexpr int NP = 2, KT = TNT_KT, NPAIR = KT / 2, NPH = (NPAIR + PPH - 1) / PPH;
  constexpr int PIECE = 32 * 128;                       // halfwords per piece image (8 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* As = reinterpret_cast<E*>(smem);                   // [2 buffers][NP][32][128]
  E* Bs = As + 2 * NP * PIECE;                          // [2 buffers][PPH pairs][NP][32][128]: a topic pair's 64 + 64 columns
  float* vtab = reinterpret_cast<float*>(Bs + 2 * PPH * NP * PIECE);   // [2 buffers][KT][32]: vbar_kn x block scale (0 for rows past the end)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave_u >> 1, wj = wave_u & 1, lr = lane & 15, lg = lane >> 4;
  // block -> (tile, topic group, split); tile fastest
  const int kgroups = (g.K + KT - 1) / KT;
  const unsigned bid = blockIdx.x;
  const int tile = (int)(bid % (unsigned)g.ntiles);
  const int gk = (int)((bid / (unsigned)g.ntiles) % (unsigned)kgroups);
  const int sp = (int)(bid / (unsigned)(g.ntiles * kgroups));
  int I = 0, J = tile;
  {
    const int nJ = (g.ncols + 63) / 64;
    for (;; ++I) { const int cnt = 2 * I + 2 < nJ ? 2 * I + 2 : nJ; if (J < cnt) break; J -= cnt; }
  }
  const int i0 = I * 128, j0 = J * 64;
  const int k0 = gk * KT, kg = min(KT, g.K - k0), npair = (kg + 1) >> 1;
  const int64_t r0 = (int64_t)sp * g.rows_per_split;
  int64_t r1 = r0 + g.rows_per_split; if (r1 > g.nrows) r1 = g.nrows;
  const int nch = r0 < r1 ? (int)((r1 - r0 + 31) / 32) : 0;
  // this wave's 32 x 32 sub-tile lies entirely above the diagonal: nothing to multiply (reduce_slabs mirrors it)
  const bool active = (i0 + 32 * wi) >= (j0 + 32 * wj);

  f32x4 acc[KT][2][2];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[k][a][b] = f32x4{0, 0, 0, 0};

  // ---- staging maps
  const unsigned as_base = lds_addr(As) + (unsigned)wave_u * 1024u;      // A: this wave moves row block `wave` (4 rows) of every piece
  auto dma_a = [&](int c) {
    const int k = 4 * wave_u + (lane >> 4);
    const int c8 = (lane & 15) ^ (tnb_code(k) << 1);
    int64_t n = r0 + (int64_t)c * 32 + k;
    n = n < g.nrows ? n : g.nrows - 1;
    const int col = (i0 + c8 * 8 < g.ncols) ? i0 + c8 * 8 : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      glds16_asm(g.Ah + p * g.a_stride + n * g.lda + col, as_base + (unsigned)((((c & 1) * NP + p) * PIECE) * 2));
  };
  const int brow = tid >> 4, bc4 = tid & 15;                              // B: row of the chunk, 4-column group of the 64
  const bool b_ok = (j0 + 4 * bc4) < g.ncols;
  auto load_b = [&](int c) -> f32x4 {                                     // no dependence on loaded values here (see gemm_tn_split_kernel)
    const int64_t n = r0 + (int64_t)c * 32 + brow;
    const int64_t nn = n < g.nrows ? n : g.nrows - 1;
    return *reinterpret_cast<const f32x4*>(g.B + nn * g.ldb + (b_ok ? j0 + 4 * bc4 : 0));
  };
  // row factors of chunk c -> vtab[c & 1]: thread t < 32 KT handles (topic t / 32, row t % 32)
  auto stage_v = [&](int c) {
    if (tid < 32 * KT) {
      const int kk = tid >> 5, row = tid & 31;
      const int64_t n = r0 + (int64_t)c * 32 + row;
      float v = 0.0f;
      if (kk < kg && n < r1) v = g.vbar[(int64_t)(k0 + kk) * g.ldk + n] * g.sc[g.sidx_v + 2 * (k0 + kk)];
      vtab[((c & 1) * KT + kk) * 32 + row] = v;
    }
  };
  // split x (4 columns of this lane's row) for the topic pair pr of chunk c into B image (buffer q, slot sl)
  // Two neighbouring elements (xa, xb) times the row factor v -> packed fp16 pairs H = (h_a, h_b) and L = (l'_a, l'_b), l' = 2^11 (x v - h):
  // six mixed-precision FMAs - h = f16(x v) from the exact product (v_fma_mixlo/hi_f16 write one half each), the residual x v - h exactly
  // in f32 (v_fma_mix_f32 with the fp16 half as its third operand), l' = f16(2048 r).  Written as asm because (a) hipcc's own code for
  // the scalar formulation is ~14 instructions per pair (multiply, convert, convert back, subtract, multiply, convert, byte permutes) -
  // this is the vector work the kernel's phases wait for - and (b) its SLP-vectorized form of it produced run-to-run different results
  // (packed-f32 code with op_sel; ROCm 7.2).  The trailing s_nop pads the VALU write -> LDS store-data read that follows in compiler code.
  const float c2048 = 2048.0f;
  auto split2 = [&](float xa, float xb, float v, unsigned& H, unsigned& L) {
    float ra, rb2;
    asm("v_fma_mixlo_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %4, 0\n\t"
        "v_fma_mix_f32 %1, %3, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %2, %5, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(H), "=&v"(ra), "=&v"(rb2) : "v"(xa), "v"(v), "v"(xb));
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %3, 0\n\t"
        "s_nop 1"
        : "=&v"(L) : "v"(ra), "v"(rb2), "v"(c2048));
  };
  // split x (4 columns of this lane's row) for the topic pair pr of chunk c into B image (buffer q, slot sl)
  auto split_pair = [&](const f32x4& x, int c, int pr, int q, int sl) {
#if defined(GDRF_DIAG) && GDRF_AK_ABLATE == 1      // timing-only: no split / image writes after the first chunk (wrong results)
    if (c > 0) return;
#endif
    const float* vt = vtab + ((c & 1) * KT + 2 * pr) * 32 + brow;
    const float okf = b_ok ? 1.0f : 0.0f;
    const float v0 = vt[0] * okf, v1 = vt[32] * okf;                      // 2 pr + 1 < KT always (KT even); unconditional LDS reads
    unsigned hw[2][NP][2];                                                 // [topic of the pair][piece][element pair]
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      split2(x[2 * h2], x[2 * h2 + 1], v0, hw[0][0][h2], hw[0][1][h2]);
      split2(x[2 * h2], x[2 * h2 + 1], v1, hw[1][0][h2], hw[1][1][h2]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int so = tnb_seg(brow, bc4 + 16 * t) * 4;
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(Bs + ((q * PPH + sl) * NP + s) * PIECE + so) = u32x2{hw[t][s][0], hw[t][s][1]};
      }
    }
  };
  // fragments (geometry of gemm_tn_split_kernel): two transposing reads of 4 rows x 16 columns each
  const int fq = lr >> 2, fp = lr & 3, fcode = ((lg & 1) << 2) | fq;
  const int fbase = (8 * lg + fq) * 32 + fp;
  auto frag = [&](const E* img, int g16) -> V8 {
    const int seg = fbase + ((g16 ^ fcode) << 2);
    const V4 lo = tr_read(img + seg * 4);
    const V4 hi = tr_read(img + (seg + 128) * 4);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto phase_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    gdrf_raw_barrier();
  };

  if (nch > 0) {
    const int nph = (npair + PPH - 1) / PPH;
    // ---- prologue: chunk 0's A image, row factors and the B images of its first phase
    dma_a(0);
    f32x4 rb = load_b(0);
    stage_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
#pragma unroll
    for (int sl = 0; sl < PPH; ++sl) if (sl < npair) split_pair(rb, 0, sl, 0, sl);
    phase_barrier();
    int q = 0;
#if defined(GDRF_DIAG) && GDRF_AK_ABLATE == 2
    V8 fbk[2][NP];
#endif
    V8 fa[2][NP], fa2[2];                    // fa2 = 2^-11 x the high piece: partner of the B operand's up-scaled low piece
    for (int c = 0; c < nch; ++c) {
      f32x4 rbn = rb;
      const bool more = c + 1 < nch;
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        if (ph < nph) {
          constexpr int dummy = 0; (void)dummy;
          const int first = PPH * ph;
          if (ph == 0) {
            if (active) {
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < NP; ++s) fa[a][s] = frag(As + ((c & 1) * NP + s) * PIECE, 2 * wi + a);
#pragma unroll
              for (int a = 0; a < 2; ++a) fa2[a] = fa[a][0] * (E)0.00048828125f;
            }
            if (more) {                        // next chunk: A image (into the buffer last read one chunk ago), f32 rows, row factors
              dma_a(c + 1);
              rbn = load_b(c + 1);
              stage_v(c + 1);
              if (nph == 1) phase_barrier();   // the factors are used by the splits below in this same phase when a chunk is one phase
            }
          }
          const bool same = ph + 1 < nph;      // the next phase belongs to this chunk
#pragma unroll
          for (int sl = 0; sl < PPH; ++sl) {
            // produce slot sl of the next phase's B images while this phase's are multiplied (all fragments of a pair up front cost 16
            // registers more than the 256 a two-waves-per-SIMD kernel has: spills, 10.1 -> 11.0 ms)
            if (same) { if (first + PPH + sl < npair) split_pair(rb, c, first + PPH + sl, q ^ 1, sl); }
            else if (more) { if (sl < npair) split_pair(rbn, c + 1, sl, q ^ 1, sl); }
            if (first + sl < NPAIR) {
              if (active && first + sl < npair) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  V8 fb[2][NP];
#if defined(GDRF_DIAG) && GDRF_AK_ABLATE == 2      // timing-only: one B fragment set per phase instead of one per topic (wrong results)
                  if (sl == 0 && t == 0) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                      for (int s = 0; s < NP; ++s) fb[b][s] = frag(Bs + ((q * PPH + sl) * NP + s) * PIECE, 4 * t + 2 * wj + b);
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                      for (int s = 0; s < NP; ++s) fbk[b][s] = fb[b][s];
                  } else {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                      for (int s = 0; s < NP; ++s) fb[b][s] = fbk[b][s];
                  }
#else
#pragma unroll
                  for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int s = 0; s < NP; ++s) fb[b][s] = frag(Bs + ((q * PPH + sl) * NP + s) * PIECE, 4 * t + 2 * wj + b);
#endif
#if defined(GDRF_DIAG) && GDRF_AK_ABLATE == 3      // timing-only: no MFMA (wrong results)
#pragma unroll
                  for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int s = 0; s < NP; ++s) asm volatile("" :: "v"(fb[b][s]));
                  continue;
#endif
#pragma unroll
                  for (int x = 0; x < SP::NPROD; ++x)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                      for (int b = 0; b < 2; ++b)
                        acc[(2 * (first + sl) + t) < KT ? 2 * (first + sl) + t : 0][a][b] =
                            SP::mma(x == 0 ? fa2[a] : fa[a][SP::pa(x)], fb[b][SP::pb(x)], acc[(2 * (first + sl) + t) < KT ? 2 * (first + sl) + t : 0][a][b]);   // x = 0: (h_a, l_b)
                }
              }
            }
          }
          if (ph + 1 == nph && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of A(c + 1) has landed
          phase_barrier();
          q ^= 1;
        }
      }
      rb = rbn;
    }
  }
  if (!active) return;
  const float una = g.sc[g.sidx_a + 1];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    if (k < kg) {
      const float un = una * g.sc[g.sidx_v + 2 * (k0 + k) + 1];
      float* out = g.slab + ((int64_t)sp * g.K + k0 + k) * (int64_t)g.ncols * g.ncols;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + 32 * wi + 16 * a + 4 * lg + r;
            const int j = j0 + 32 * wj + 16 * b + lr;
            if (i < g.ncols && j < g.ncols) out[(int64_t)i * g.ncols + j] = acc[k][a][b][r] * un;
          }
    }
  }
}

}  // namespace gdrf
