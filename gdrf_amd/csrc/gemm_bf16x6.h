// The f32 GEMM-shaped contractions of the step on the bf16 matrix path with exact-split emulation ("bf16x6").
//
// gfx950 runs f32-input MFMA at 1/16 of the bf16 rate (MI355X_MICROARCH.md: 157 TF vs ~2.5 PF), and ~90 % of the step is
// bound by it.  Here every f32 operand x is written as x = x1 + x2 + x3 with bf16 pieces (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 3 x 8 significand bits, residual <= 2^-24 |x|), and a product a*b is the sum of the SIX cross
// terms of weight >= 2^-16:  a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1).  The dropped terms are <= 2^-24 |ab|, i.e. the
// rounding of one f32 product, so the result has f32-level error while running on mfma_f32_16x16x32_bf16 with f32
// accumulation (16/6 = 2.7x the f32 MFMA rate).  Measured against fp64 products of the same inputs the emulated kernels are
// as close as, or closer than, the native f32 MFMA ones (tests/test_gpu_parity.py::test_bf16x6_against_fp64_product...).
//
//   split3_kernel            W, B_k = S_k S_k^T and S_k^T -> 3 bf16 pieces each, once per step
//   bwd_wbar_bf16x6_kernel   Wbar = sum_k diag(2 vbar_k) W B_k + locbar^T U - 2 diag(asum) W   (replaces gemm_nt<BwdWbarProb>)
//   fwd_t_bf16x6_2g_kernel   tt[k][n] = |S_k^T w_n|^2                                           (replaces gemm_nt<FwdTProb>)
//   gemm_tn_bf16x6_kernel    A_k = W^T diag(vbar_k) W  and  GT = W^T Wbar                        (replaces gemm_tn_kernel<float>)
//
// All keep the tiling, block maps, deterministic slab reduction and epilogues of the f32 forms they replace.
#pragma once
#include "common.h"
#include "gemm_nt.h"

namespace gdrf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Bf16x6Cfg {
  static constexpr int BK = 32;                 // f32 reduction indices per chunk = one 16x16x32 MFMA deep
  static constexpr int LDH = BK;                // LDS row = 32 halfwords = 64 bytes = four 16-byte quads, no padding
  static constexpr int PIECE = GDRF_TILE * LDH; // halfwords per piece image of a 128-row operand tile
  static constexpr int LDS_BYTES = 2 * 3 * PIECE * 2;    // A and B, 3 pieces each: 49152
};
// LDS image of an operand tile: element (row, k) lives at halfword  row*32 + ((k>>3) ^ swz(row))*8 + (k&7), swz(row) = 3 if
// (row & 8) else 0.  ds_read_b128 serves a wave in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... with bank =
// dword mod 64 (MI355X_MICROARCH.md, LDS): an MFMA fragment read (lane -> row lane&15, quad lane>>4) then touches, in every
// group, all 16 (row mod 4, physical quad) pairs exactly once - conflict-free; a padded 80-byte row is not (measured: 47 %
// of the LDS cycles were bank conflicts).  ds_write_b128 (8 contiguous lanes = 2 whole rows = 32 distinct banks) is too.
__device__ __forceinline__ int bf16x6_swz(int row) { return (row & 8) ? 3 : 0; }
__device__ __forceinline__ int bf16x6_off(int row, int k) { return row * 32 + (((k >> 3) ^ bf16x6_swz(row)) << 3) + (k & 7); }

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// out[p * stride + i] = p-th bf16 piece of in[i]; 4 elements per thread (n multiple of 4)
__global__ void split3_kernel(const float* __restrict__ in, int64_t n, __bf16* __restrict__ out, int64_t stride) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const f32x4 x = *reinterpret_cast<const f32x4*>(in + i);
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(x[e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
  *reinterpret_cast<bf16x4*>(out + i) = h;
  *reinterpret_cast<bf16x4*>(out + stride + i) = m;
  *reinterpret_cast<bf16x4*>(out + 2 * stride + i) = l;
}

// the same pieces in k-blocked order for the NT kernels' small operands (B_k, S_k^T):  in[b][r][c] (b < nb, r < R, c < C, C a
// multiple of 32)  ->  out[p][b][c / 32][r][c % 32].  A tile's 32-wide reduction chunk is then ONE contiguous block of whole
// 128-byte lines (rows x 64 B), instead of 64 B out of every 1 KB row: the row-major form left the load path saturated
// (measured: ~4000 cycles per prefetched load, 1500 of every 4600 cycles per chunk spent waiting for them).
__global__ void split3_blocked_kernel(const float* __restrict__ in, int64_t n, int R, int C, __bf16* __restrict__ out, int64_t stride) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const f32x4 x = *reinterpret_cast<const f32x4*>(in + i);
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(x[e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
  const int64_t rc = (int64_t)R * C, b = i / rc, rem = i - b * rc;
  const int r = (int)(rem / C), c = (int)(rem - (int64_t)r * C);
  const int64_t o = b * rc + ((int64_t)(c >> 5) * R + r) * 32 + (c & 31);
  *reinterpret_cast<bf16x4*>(out + o) = h;
  *reinterpret_cast<bf16x4*>(out + stride + o) = m;
  *reinterpret_cast<bf16x4*>(out + 2 * stride + o) = l;
}

struct BwdWbarBf16Args {
  const float* W; const __bf16* Wh; int64_t w_stride;   // W (f32, epilogue) and its 3 bf16 pieces [p][row][Mp]
  int64_t nrows; int M, Mp, K;
  const __bf16* Bh; int64_t piece_stride;     // Bh[p][k][i / 32][col][i % 32] (k-blocked), piece_stride = K*Mp*Mp
  const float* vbar; const float* locbar; int64_t ldk;
  const float* asum; const float* U; float* Wbar;
};

// NG = 1: one 256-thread workgroup per tile, two workgroups per CU (they share the CU's SIMDs at random).
// NG = 2: one 512-thread workgroup per CU holding TWO such tiles, one per wave group (waves 0-3 / 4-7: a workgroup's waves
// go to the SIMDs cyclically, so every SIMD hosts one wave of each group), run phase-shifted on purpose: in every phase one
// group multiplies its staged chunk while the other moves its next chunk from registers to LDS and issues the loads of the
// one after, then they swap.  Measured cause (s_memtime per wave, NG = 1): the multiply phase is 1810 cycles (MFMA 1536) but
// every wave then waits ~1500 cycles at the barrier - each SIMD's two waves contend for the matrix pipe at random and the
// barrier paces a workgroup by its unluckiest wave.  With the groups alternating, the multiplying wave has its SIMD's pipe
// to itself and the staging hides behind it.  The phase barrier is a raw s_barrier after s_waitcnt lgkmcnt(0): a
// __syncthreads() would also drain vmcnt and wait for the prefetch just issued.
template <int NG>
__global__ __launch_bounds__(256 * NG, 2) void bwd_wbar_bf16x6_kernel(BwdWbarBf16Args g) {
  using CF = Bf16x6Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int gp = NG == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  // LDS: NG = 1: [A][B][scale];  NG = 2: [A of group 0][A of group 1][B buffer 0][B buffer 1][scale 0][scale 1] - the two
  // groups work on the same column tile, so the B chunk is staged ONCE (by group 1) into a double-buffered image both read
  constexpr int IMG = 3 * CF::PIECE;                          // halfwords per operand image
  const size_t tab = ((size_t)g.K * GDRF_TILE * sizeof(float) + 15) & ~(size_t)15;
  __bf16* As = reinterpret_cast<__bf16*>(smem) + (NG == 2 ? gp * IMG : 0);              // [3][128][LDH]
  __bf16* Bs = reinterpret_cast<__bf16*>(smem) + (NG == 2 ? 2 * IMG : IMG);             // [NG][3][128][LDH]
  float* scaleS = reinterpret_cast<float*>(smem + (size_t)(NG == 2 ? 4 : 2) * IMG * 2 + (size_t)gp * tab);  // [K][128]  2 vbar_kn

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int nct = (g.Mp + GDRF_TILE - 1) / GDRF_TILE;
  int64_t rtile; int ct;
  if constexpr (NG == 1) {
    NTXcdMap{}.map_block(blockIdx.x, nct, false, rtile, ct);
  } else {        // the same XCD <-> column tile affinity; the two groups take adjacent row tiles of one column tile
    const unsigned w = blockIdx.x;
    if (nct <= 8 && (8 % nct) == 0 && (gridDim.x % 8u) == 0) {
      const unsigned xcd = w & 7u, idx = w >> 3, per = 8u / (unsigned)nct;
      ct = (int)(xcd % (unsigned)nct);
      rtile = 2 * ((int64_t)idx * per + xcd / (unsigned)nct) + gp;
    } else {
      ct = (int)(w % (unsigned)nct);
      rtile = 2 * (int64_t)(w / (unsigned)nct) + gp;
    }
  }
  const int64_t m0 = rtile * GDRF_TILE;
  const int n0 = ct * GDRF_TILE;
  const int K = g.K, Mp = g.Mp;

  for (int e = tid; e < K * GDRF_TILE; e += 256) {
    const int k = e / GDRF_TILE, r = e - k * GDRF_TILE;
    scaleS[e] = (m0 + r < g.nrows) ? 2.0f * g.vbar[(int64_t)k * g.ldk + m0 + r] : 0.0f;
  }
  // staging map, both operands: per piece 2 x 16-byte vectors of 8 bf16 per thread (vector v = tid + 256 j: row v>>2,
  // k offset 8*(v&3))
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};

  bf16x8 ra[3][2], rb[3][2];
  auto load_a = [&](int kA) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, kq = (v & 3) * 8;
      int64_t row = m0 + (v >> 2);
      row = row < g.nrows ? row : 0;          // rows past the end read row 0; their output rows are never stored
#pragma unroll
      for (int p = 0; p < 3; ++p) ra[p][j] = *reinterpret_cast<const bf16x8*>(g.Wh + p * g.w_stride + row * Mp + kA + kq);
    }
  };
  auto load_b = [&](int kA, int rep) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, row = v >> 2, kq = (v & 3) * 8;
      const int col = (n0 + row < Mp) ? n0 + row : 0;     // likewise: columns >= Mp are never stored
#pragma unroll
      for (int p = 0; p < 3; ++p)
        rb[p][j] = *reinterpret_cast<const bf16x8*>(g.Bh + p * g.piece_stride + (((int64_t)rep * (Mp >> 5) + (kA >> 5)) * Mp + col) * 32 + kq);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, off = bf16x6_off(v >> 2, (v & 3) * 8);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(As + p * CF::PIECE + off) = ra[p][j];
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, off = bf16x6_off(v >> 2, (v & 3) * 8);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(Bs + (buf * 3 + p) * CF::PIECE + off) = rb[p][j];
    }
  };
  // one staged chunk: P = A B^T through the six cross products (small terms first), then acc += diag(scale) P.
  // The A fragments of a chunk serve all K topic reps, so they are read from LDS once (rep 0) and stay in registers; the
  // B fragments are read one 16-column group at a time.  LDS reads per chunk: 12 fragments instead of 24.
  bf16x8 fa[4][3];
  const int frag = lr * 32 + ((lg ^ bf16x6_swz(lr)) << 3);      // this lane's fragment offset inside a 16-row group
  auto read_a = [&]() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[a][p] = *reinterpret_cast<const bf16x8*>(As + p * CF::PIECE + (wr * 64 + a * 16) * 32 + frag);
  };
  auto read_b = [&](bf16x8 (&fb)[3], int b, int buf) {
#pragma unroll
    for (int p = 0; p < 3; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(Bs + (buf * 3 + p) * CF::PIECE + (wc * 64 + b * 16) * 32 + frag);
  };
  auto compute = [&](const float* sc_row /* scaleS + rep*128, or nullptr for scale 1 */, int buf) {
    f32x4 s4[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
      s4[a] = sc_row ? *reinterpret_cast<const f32x4*>(sc_row + wr * 64 + a * 16 + lg * 4) : f32x4{1, 1, 1, 1};   // rows 4*lg + r
    bf16x8 fbq[2][3];                      // the next column group's fragments are read while this one multiplies
    read_b(fbq[0], 0, buf);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bf16x8 (&fb)[3] = fbq[b & 1];
      if (b + 1 < 4) read_b(fbq[(b + 1) & 1], b + 1, buf);
      f32x4 P[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[2], f32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][1], fb[1], P[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][2], fb[0], P[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[1], P[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][1], fb[0], P[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) P[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[0], P[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += s4[a][r] * P[a][r];
    }
  };

  const int nk = Mp / CF::BK, nchunks = nk * K;
  if constexpr (NG == 1) {
    load_a(0);
    load_b(0, 0);
    for (int c = 0; c < nchunks; ++c) {
      const int q = c / K, rep = c - q * K;
      __syncthreads();
      if (rep == 0) store_a();
      store_b(0);
      __syncthreads();
      if (c + 1 < nchunks) {
        const int q1 = (c + 1) / K, rep1 = (c + 1) - q1 * K;
        if (rep1 == 0) load_a(q1 * CF::BK);
        load_b(q1 * CF::BK, rep1);
      }
      if (rep == 0) read_a();
      compute(scaleS + rep * GDRF_TILE, 0);
    }
  } else {
    // Staging by LDS-DMA (global_load_lds_dwordx4: no destination registers, no ds_write - a staging wave's ds_write_b128
    // were measured starving, ~1600 cycles for six, while its SIMD's other wave streams MFMAs).  One wave-instruction moves a
    // 16-row block of one piece image, 1 KB: lane l lands at block + 16 l = row l>>2, physical quad l&3, so its SOURCE is
    // logical quad (l&3) ^ swz(row) - the image swizzle goes on the global address (cdna_hip_programming.md 5.4 rule 21).
    const int drow = lane >> 2, dq = ((lane & 3) ^ bf16x6_swz(drow)) * 8;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform: the DMA's LDS base goes to M0 without a waterfall loop
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma_b = [&](int c) {             // B chunk c -> buffer c & 1; this wave moves row blocks wave and wave + 4 of each piece
      if (c >= nchunks) return;
      const int q = c / K, rep = c - q * K;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rbk = wave_u + 4 * i, row = rbk * 16 + drow;
        const int col = (n0 + row < Mp) ? n0 + row : 0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          __builtin_amdgcn_global_load_lds((gptr_t)(g.Bh + p * g.piece_stride + (((int64_t)rep * (Mp >> 5) + q) * Mp + col) * 32 + dq),
                                           (lptr_t)(Bs + ((c & 1) * 3 + p) * CF::PIECE + rbk * 512), 16, 0, 0);
      }
    };
    auto dma_a = [&](int q) {             // this group's A image for reduction block q
      if (q >= nk) return;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rbk = wave_u + 4 * i;
        int64_t row = m0 + rbk * 16 + drow;
        row = row < g.nrows ? row : 0;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          __builtin_amdgcn_global_load_lds((gptr_t)(g.Wh + p * g.w_stride + row * Mp + q * CF::BK + dq),
                                           (lptr_t)(As + p * CF::PIECE + rbk * 512), 16, 0, 0);
      }
    };
    auto mult = [&](int c) {
      const int q = c / K, rep = c - q * K;
      if (rep == 0) read_a();
      compute(scaleS + rep * GDRF_TILE, c & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the DMAs this group issued one phase ago have had a whole phase
    };
    auto phase_barrier = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's LDS reads are done; a DMA just issued stays in flight
      __builtin_amdgcn_s_barrier();
    };
    // Chunk t is multiplied by group 0 in phase 2t and by group 1 in phase 2t+1, both from B buffer t & 1 and their own A
    // image (whose fragments then stay in registers for the K topic reps).  DMAs are issued in a group's idle phase, land
    // during its next multiply phase, are retired by the vmcnt(0) that ends it, and are first read a barrier later:
    //   group 1, phase 2u   : B chunk u+1 -> buffer (u+1)&1 (last read in phase 2u-1); its own A if chunk u+1 opens a block
    //   group 0, phase 2u+1 : its own A if chunk u+2 opens a block (needed in phase 2u+4; image last read >= 2 phases ago, K >= 2)
    dma_a(0);
    if (gp == 1) dma_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    for (int ph = 0; ph < 2 * nchunks; ++ph) {
      const int t = ph >> 1;
      if ((ph & 1) == gp) {
        mult(t);
        phase_barrier();
        } else if (gp == 0) {
        const int c = t + 2;
        if (c < nchunks && c % K == 0) dma_a(c / K);
          phase_barrier();
        } else {
        const int c = t + 1;
        dma_b(c);
        if (c < nchunks && c % K == 0) dma_a(c / K);
          phase_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // rank-K epilogue term locbar^T U as extra chunk(s), split on the fly
  for (int x = 0; x * CF::BK < K; ++x) {
    __syncthreads();
    for (int e = tid; e < GDRF_TILE * CF::BK; e += 256) {
      const int kk = e / GDRF_TILE, row = e - kk * GDRF_TILE, k = x * CF::BK + kk;
      const float va = (k < K && m0 + row < g.nrows) ? g.locbar[(int64_t)k * g.ldk + m0 + row] : 0.0f;
      const float vb = (k < K && n0 + row < g.M) ? g.U[(int64_t)k * g.M + n0 + row] : 0.0f;
      __bf16 h, m, l;
      split3(va, h, m, l);
      const int o = bf16x6_off(row, kk);
      As[o] = h; As[CF::PIECE + o] = m; As[2 * CF::PIECE + o] = l;
      split3(vb, h, m, l);
      Bs[o] = h; Bs[CF::PIECE + o] = m; Bs[2 * CF::PIECE + o] = l;
    }
    __syncthreads();
    read_a();
    compute(nullptr, 0);
  }
  // Wbar = acc - 2 asum W.  The MFMA accumulator layout (a lane owns one column of four rows) would make this 64 scalar
  // loads of W and 64 scalar stores per lane in a dependent sequence - measured ~300k cycles per workgroup, a third of its
  // lifetime.  Instead each wave transposes its 64 x 64 quadrant through a private LDS tile, 32 rows at a time, and moves whole
  // 256-byte row segments: 16 float4 loads and 16 float4 stores per lane.
  __syncthreads();                                           // every wave is done with the operand images
  {
    constexpr int TS = 68;                                   // tile row stride in floats (272 B: 16-byte aligned, bank-shifted)
    float* tile = reinterpret_cast<float*>(smem) + (size_t)((NG == 2 ? gp * 4 : 0) + wave) * (32 * TS);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[(a2 * 16 + lg * 4 + r) * TS + b * 16 + lr] = acc[2 * h + a2][b][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave writes and reads: LDS is in order per wave
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = (lane >> 4) + 4 * i, cv = (lane & 15) * 4;
        const int64_t m = m0 + wr * 64 + h * 32 + rr;
        const int n = n0 + wc * 64 + cv;
        const f32x4 t = *reinterpret_cast<const f32x4*>(tile + rr * TS + cv);
        if (m < g.nrows && n < Mp) {                         // Mp is a multiple of 32: a float4 never straddles the edge
          const float as2 = 2.0f * g.asum[m];
          const f32x4 w = *reinterpret_cast<const f32x4*>(g.W + m * Mp + n);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = t[e] - as2 * w[e];
          *reinterpret_cast<f32x4*>(g.Wbar + m * Mp + n) = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the tile is overwritten with the other half
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// tt[k][n] = || S_k^T w_n ||^2 on the same emulation: T_k = W S_k lives only in the accumulators; a row tile and topic walk
// the column tiles (triangular k range i >= j), as gemm_nt<FwdTProb> does.
struct FwdTBf16Args {
  const __bf16* Wh; int64_t w_stride; int64_t nrows; int Mp, K;
  int KG; int rt8;                            // topics per group (see the block map), ceil(row-tile pairs / 8)
  const __bf16* STh; int64_t piece_stride;    // STh[p][k][i / 32][j][i % 32] = pieces of S_k[i][j] (k-blocked)
  float* tt; int64_t ldt;
};

// ------------------------------------------------------------------------------------------------------------------
// tt in the two-group LDS-DMA structure of bwd_wbar_bf16x6_kernel<2>: a 512-thread workgroup holds two adjacent row tiles of
// one topic, one per wave group; both walk the same (column tile, reduction chunk) sequence, so the S_k^T chunk is staged once
// (by group 1) into a double-buffered image both read, and every group double-buffers its own W chunk.  Chunk c is multiplied
// by group 0 in phase 2c and by group 1 in phase 2c+1; DMAs are issued in a group's idle phase two chunks ahead (group 0) / one
// chunk ahead (group 1), land during its next multiply phase and are retired by the vmcnt(0) that ends it.
// Grid: 8 * K * rt8 with rt8 = ceil(row-tile pairs / 8).  Block map: XCD = blockIdx & 7 owns the pairs r * 8 + xcd; topics go
// in groups of KG, group-major (every XCD first runs all its pairs for topics [0, KG), then [KG, 2 KG), ...) so that only KG
// topics' S^T pieces (~1 MB each at M = 512) are live in its 4 MB L2 at a time; the KG workgroups of one pair are adjacent.
__global__ __launch_bounds__(512, 2) void fwd_t_bf16x6_2g_kernel(FwdTBf16Args g) {
  using CF = Bf16x6Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = 3 * CF::PIECE;                          // halfwords per operand image
  const int gp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  __bf16* As = reinterpret_cast<__bf16*>(smem) + gp * 2 * IMG;      // [2 buffers][3][128][32] of this group
  __bf16* Bs = reinterpret_cast<__bf16*>(smem) + 4 * IMG;           // [2 buffers][3][128][32] shared
  const int Mp = g.Mp;
  const int nct = (Mp + GDRF_TILE - 1) / GDRF_TILE;
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned per_group = (unsigned)g.KG * (unsigned)g.rt8;
  const int grp = (int)(idx / per_group);
  const unsigned rem = idx - (unsigned)grp * per_group;
  const int kg = min(g.KG, g.K - grp * g.KG);
  const int64_t rtile = 2 * ((int64_t)(rem / (unsigned)kg) * 8 + xcd) + gp;
  const int bz = grp * g.KG + (int)(rem % (unsigned)kg);
  const int64_t m0 = rtile * GDRF_TILE;                        // a tile past the end runs on clamped rows; its tt is never stored

  int nch = 0;                                                 // chunks of the triangular walk: column tile ct covers k in [128 ct, Mp)
  for (int ct = 0; ct < nct; ++ct) nch += (Mp - ct * GDRF_TILE) / CF::BK;
  auto decode = [&](int c, int& ct, int& kA, bool& first, bool& last) {
    for (ct = 0;; ++ct) {
      const int len = (Mp - ct * GDRF_TILE) / CF::BK;
      if (c < len) { kA = ct * GDRF_TILE + c * CF::BK; first = c == 0; last = c == len - 1; return; }
      c -= len;
    }
  };
  const int drow = lane >> 2, dq = ((lane & 3) ^ bf16x6_swz(drow)) * 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably wave-uniform: the DMA's LDS base goes to M0 without a waterfall loop
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_b = [&](int c) {
    if (c >= nch) return;
    int ct, kA; bool f, l;
    decode(c, ct, kA, f, l);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbk = wave_u + 4 * i, row = rbk * 16 + drow;
      const int col = (ct * GDRF_TILE + row < Mp) ? ct * GDRF_TILE + row : 0;      // columns >= Mp are skipped when the tile is folded
#pragma unroll
      for (int p = 0; p < 3; ++p)
        __builtin_amdgcn_global_load_lds((gptr_t)(g.STh + p * g.piece_stride + (((int64_t)bz * (Mp >> 5) + (kA >> 5)) * Mp + col) * 32 + dq),
                                         (lptr_t)(Bs + ((c & 1) * 3 + p) * CF::PIECE + rbk * 512), 16, 0, 0);
    }
  };
  auto dma_a = [&](int c) {
    if (c >= nch) return;
    int ct, kA; bool f, l;
    decode(c, ct, kA, f, l);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbk = wave_u + 4 * i;
      int64_t row = m0 + rbk * 16 + drow;
      row = row < g.nrows ? row : 0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        __builtin_amdgcn_global_load_lds((gptr_t)(g.Wh + p * g.w_stride + row * Mp + kA + dq),
                                         (lptr_t)(As + ((c & 1) * 3 + p) * CF::PIECE + rbk * 512), 16, 0, 0);
    }
  };
  float rs[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[a][r] = 0;
  f32x4 acc[4][4];
  const int frag = lr * 32 + ((lg ^ bf16x6_swz(lr)) << 3);
  auto mult = [&](int c) {
    int ct, kA; bool first, last;
    decode(c, ct, kA, first, last);
    if (first) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    }
    const __bf16* Ab = As + (c & 1) * IMG;
    const __bf16* Bb = Bs + (c & 1) * IMG;
    bf16x8 fb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[p][b] = *reinterpret_cast<const bf16x8*>(Bb + p * CF::PIECE + (wc * 64 + b * 16) * 32 + frag);
    bf16x8 faq[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) faq[0][p] = *reinterpret_cast<const bf16x8*>(Ab + p * CF::PIECE + (wr * 64) * 32 + frag);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 (&fa)[3] = faq[a & 1];
      if (a + 1 < 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p) faq[(a + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(Ab + p * CF::PIECE + (wr * 64 + (a + 1) * 16) * 32 + frag);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0][b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0][b], acc[a][b], 0, 0, 0);
    }
    if (last) {                                                 // fold the finished column tile into the row sums
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool cok = ct * GDRF_TILE + wc * 64 + b * 16 + lr < Mp;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) rs[a][r] += cok ? acc[a][b][r] * acc[a][b][r] : 0.0f;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto phase_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  dma_a(0);
  if (gp == 0) dma_a(1); else dma_b(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  phase_barrier();
  for (int ph = 0; ph < 2 * nch; ++ph) {
    const int t = ph >> 1;
    if ((ph & 1) == gp) {
      mult(t);
    } else if (gp == 0) {
      dma_a(t + 2);                        // phase 2t+1: buffer (t+2)&1 was last read in phase 2t
    } else {
      dma_a(t + 1);                        // phase 2t: own buffer (t+1)&1 last read in phase 2t-1; so was the shared B buffer
      dma_b(t + 1);
    }
    phase_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // row sums: 16-lane groups, then the two waves of a group that share the rows, through LDS
  float* rsum = reinterpret_cast<float*>(smem) + gp * GDRF_TILE;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[a][r] = group16_sum(rs[a][r]);
  __syncthreads();
  if (wc == 0 && lr == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) rsum[wr * 64 + a * 16 + lg * 4 + r] = rs[a][r];
  }
  __syncthreads();
  if (wc == 1 && lr == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) rsum[wr * 64 + a * 16 + lg * 4 + r] += rs[a][r];
  }
  __syncthreads();
  if (tid < GDRF_TILE) {
    const int64_t m = m0 + tid;
    if (m < g.nrows) g.tt[(int64_t)bz * g.ldt + m] = rsum[tid];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TN form (reduction over observations):  C[i][j] = sum_n A[n][i] * s[n] * B[n][j]  on the same emulation; replaces
// gemm_tn_kernel<float> for A_k = W^T diag(vbar_k) W (sym) and GT = W^T Wbar.  A comes pre-split (Wh), B is f32 and is
// scaled by s[n] and split when it is staged (the scale runs along the reduction index, so it cannot be pulled out of the
// product).  Both operands are staged as they lie in memory (n-major rows) and the MFMA fragments (8 consecutive n per lane)
// are gathered with the transposing LDS read ds_read_b64_tr_b16 (cdna_hip_programming.md T10).
//
// LDS piece image of a 32 (n) x 128 (column) chunk: row-major (256-byte rows, so an LDS-DMA instruction fills 4 whole rows
// from the row-major pieces), with the 8-byte column quad c4 XOR-swizzled by the row:
//   seg(k, c4) = k * 32 + (c4 ^ (code(k) << 2)),  code(k) = ((k >> 3) & 1) << 2 | (k & 3)      (8-byte segments)
// A transposing fragment read of a half-wave touches rows k = 8 lg + 4 h + q (lg in {0,1} or {2,3}, q = 0..3) x quads base + p,
// p = 0..3: the eight (lg & 1, q) codes send the eight rows to eight different 32-byte groups of the 256-byte bank window -
// conflict-free.  The pre-split A operand is staged by DMA (double-buffered, no registers, no LDS stores); the B operand, which
// must be scaled and split, goes through registers and 6 ds_write_b128 per thread.
struct TNBf16Args {
  const __bf16* Ah; int64_t a_stride; int64_t lda;     // pieces [p][n][lda]
  const float* B; int64_t ldb;                         // [n][ldb]
  const float* scale; int64_t scale_bs;                // optional per-row scale, batch stride; nullptr = 1
  int64_t nrows, rows_per_split;                       // rows_per_split multiple of 32
  int ncols, sym;
  float* slab;                                         // [nsplit][nbatch][ncols][ncols]
  int nbatch, nsplit;
};

__device__ __forceinline__ int tnb_code(int k) { return (((k >> 3) & 1) << 2) | (k & 3); }
__device__ __forceinline__ int tnb_seg(int k, int c4) { return k * 32 + (c4 ^ (tnb_code(k) << 2)); }

// 192 registers at most (amdgpu_num_vgpr counts half of the unified file): one wave of this kernel then shares a SIMD's 512 with two
// 160-register waves of bwd_knm on the other stream (api.hip), which is how G^T runs inside that kernel's stalls
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(96))) void gemm_tn_bf16x6_kernel(TNBf16Args g) {
  constexpr int PIECE = 32 * 128;             // halfwords per piece image (8 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);           // [2 buffers][3][32][128]
  __bf16* Bs = As + 6 * PIECE;                             // [3][32][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int nt = (g.ncols + GDRF_TILE - 1) / GDRF_TILE;
  const int ntiles = g.sym ? nt * (nt + 1) / 2 : nt * nt;
  int tile, b, sp;                            // block -> (tile, batch, split) exactly as gemm_tn_kernel (gemm_tn.h)
  {
    const unsigned bid = blockIdx.x;
    if ((g.nsplit & 7) == 0) {
      const unsigned xcd = bid & 7u, idx = bid >> 3, per = (unsigned)(ntiles * g.nbatch);
      const unsigned sl = idx / per, r = idx - sl * per;
      sp = (int)(sl * 8u + xcd); b = (int)(r / (unsigned)ntiles); tile = (int)(r % (unsigned)ntiles);
    } else {
      tile = (int)(bid % (unsigned)ntiles);
      const unsigned r = bid / (unsigned)ntiles;
      b = (int)(r % (unsigned)g.nbatch); sp = (int)(r / (unsigned)g.nbatch);
    }
  }
  int ti, tj;
  if (g.sym) {
    int t = tile; ti = 0;
    while (t >= ti + 1) { t -= ti + 1; ++ti; }
    tj = t;
  } else { ti = tile / nt; tj = tile % nt; }
  const int i0 = ti * GDRF_TILE, j0 = tj * GDRF_TILE;
  const int64_t r0 = (int64_t)sp * g.rows_per_split;
  int64_t r1 = r0 + g.rows_per_split; if (r1 > g.nrows) r1 = g.nrows;
  const float* sc = g.scale ? g.scale + (int64_t)b * g.scale_bs : nullptr;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0, 0, 0, 0};

  // A: DMA, 4 rows (1 KB) per instruction: lane l -> row l>>4, physical 16-byte unit l&15 = logical unit (l&15) ^ (code << 1);
  // this wave moves row blocks wave and wave + 4 of every piece.  Rows past the split / the end meet a zero B row; columns past
  // the end give rows of C that are never stored.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // provably wave-uniform: the DMA's LDS base goes to M0 without a waterfall loop
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int nch = r0 < r1 ? (int)((r1 - r0 + 31) / 32) : 0;
  auto dma_a = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = wave_u + 4 * i, k = 4 * blk + (lane >> 4);
      const int c8 = (lane & 15) ^ (tnb_code(k) << 1);
      int64_t n = r0 + (int64_t)c * 32 + k;
      n = n < g.nrows ? n : g.nrows - 1;
      const int col = (i0 + c8 * 8 < g.ncols) ? i0 + c8 * 8 : 0;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        __builtin_amdgcn_global_load_lds((gptr_t)(g.Ah + p * g.a_stride + n * g.lda + col),
                                         (lptr_t)(As + ((c & 1) * 3 + p) * PIECE + blk * 512), 16, 0, 0);
    }
  };
  // B: 8 columns of a row per vector pair (q = tid&3 -> row 4*(khi + 4 i) + q, c8 = (tid>>2)&15, khi = tid>>6), 2 rows per thread:
  // two adjacent f32x4 loads, and after the split ONE ds_write_b128 per piece (the swizzle keeps the quads 2 c8, 2 c8 + 1 adjacent)
  const int sq = tid & 3, b_c8 = (tid >> 2) & 15, b_kh = tid >> 6;
  const bool b_ok = (j0 + b_c8 * 8) < g.ncols;                // ncols multiple of 32
  f32x4 rb[4];                                                 // [2 i + half]
  float rs[2];
  auto load_b = [&](int c) {
    const int64_t rbase = r0 + (int64_t)c * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t n = rbase + 4 * (b_kh + 4 * i) + sq;
      const bool ok = b_ok && n < r1;
      const float* src = g.B + (ok ? n : r0) * g.ldb + (b_ok ? j0 + b_c8 * 8 : 0);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
      rb[2 * i] = ok ? v0 : f32x4{0, 0, 0, 0};
      rb[2 * i + 1] = ok ? v1 : f32x4{0, 0, 0, 0};
      rs[i] = (sc && ok) ? sc[n] : 1.0f;
    }
  };
  // fragments: two transposing reads (k = 8 lg + q and + 4) of 4 rows x 16 columns each; seg = (8 lg + 4 h + q) * 32 + ((4 w + t) ^ code) * 4 + p
  const int fq = lr >> 2, fp = lr & 3, fcode = ((lg & 1) << 2) | fq;
  const int fbase = (8 * lg + fq) * 32 + fp;
  int xa[4], xb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { xa[t] = fbase + (((4 * wr + t) ^ fcode) << 2); xb[t] = fbase + (((4 * wc + t) ^ fcode) << 2); }
  auto frag = [&](const __bf16* img, int seg) -> bf16x8 {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + seg * 4));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + (seg + 128) * 4));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  // symmetric problem, diagonal tile: the quadrant above the diagonal (rows 0-63 x columns 64-127) is the mirror image of the one
  // below it and reduce_slabs_kernel takes it from there.  Its wave skips fragment reads and MFMAs - a quarter of the tile's LDS
  // read traffic, which is what bounds this kernel - and only stages.
  const bool idle = g.sym && ti == tj && __builtin_amdgcn_readfirstlane(wave) == 1;
  if (nch > 0) { dma_a(0); load_b(0); }
  for (int c = 0; c < nch; ++c) {
    // scale and split the prefetched B vectors BEFORE the barrier: the conversion then overlaps the other waves' MFMAs
    // instead of sitting in the barrier-to-barrier staging section
    bf16x8 hb[2], mb[2], lb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 x1, x2, x3;
        split3(rb[2 * i + (e >> 2)][e & 3] * rs[i], x1, x2, x3);
        hb[i][e] = x1; mb[i][e] = x2; lb[i][e] = x3;
      }
    __syncthreads();                       // (its vmcnt(0) also retires this wave's DMA of chunk c, issued one iteration ago)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int so = tnb_seg(4 * (b_kh + 4 * i) + sq, 2 * b_c8) * 4;
      *reinterpret_cast<bf16x8*>(Bs + so) = hb[i];
      *reinterpret_cast<bf16x8*>(Bs + PIECE + so) = mb[i];
      *reinterpret_cast<bf16x8*>(Bs + 2 * PIECE + so) = lb[i];
    }
    __syncthreads();
    if (c + 1 < nch) { dma_a(c + 1); load_b(c + 1); }          // A buffer (c+1)&1 was last read in iteration c-1
    if (idle) continue;                                        // staged and synchronised with the others; nothing of its own to multiply
    const __bf16* Ab = As + (c & 1) * 3 * PIECE;
    bf16x8 fb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) fb[p][t] = frag(Bs + p * PIECE, xb[t]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[p] = frag(Ab + p * PIECE, xa[a]);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2][t], acc[a][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1][t], acc[a][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0][t], acc[a][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1][t], acc[a][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0][t], acc[a][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0][t], acc[a][t], 0, 0, 0);
    }
  }
  if (idle) return;
  float* out = g.slab + ((int64_t)sp * g.nbatch + b) * (int64_t)g.ncols * g.ncols;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 64 + a * 16 + lg * 4 + r;
        const int j = j0 + wc * 64 + c * 16 + lr;
        if (i < g.ncols && j < g.ncols) out[(int64_t)i * g.ncols + j] = acc[a][c][r];
      }
}

}  // namespace gdrf
