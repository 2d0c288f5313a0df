// tt_k[n] = || (W S_k^T)[n, :] ||^2 on the split-fp16 matrix path - the ONE-WAVE-PER-SIMD form of fwd_t_split_q4_kernel (gemm_split.h).
//
// What the 16-wave form waits for is ISSUE, not the matrix pipe (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost'; measured on
// its sibling bwd_wbar_f16_k64_kernel: 9.6 ms with every request, LDS read and update ablated against 13.3 with them, for 7.6 ms of MFMA
// at the held clock): per SIMD and 32-deep chunk four waves issue 192 MFMAs (8 issue cycles each), 64 ds_read_b128 and 16 LDS-DMA
// requests at 60 - 185 cycles apiece, behind a 16-wave barrier; 8.2 ms for 3.8 ms of MFMAs.
//
// Here a workgroup is 4 waves, one per SIMD, and a wave owns 64 rows x 256 columns - 64 accumulator tiles, all 256 AGPRs - so that
//   * all four waves walk the same columns: the triangle (S_k^T is zero above the diagonal: column j needs the reduction indices >= j
//     only) is cut at 16-column granularity, uniformly for the workgroup - a chunk multiplies the column tiles j0 <= kA + 31 and requests
//     only those rows of the S_k^T image (the 16-wave form cuts at 64 / 128 columns: 25 % more MFMAs at M = 512);
//   * per chunk a wave reads 8 A and up to 32 B fragments for up to 192 MFMAs, and the workgroup issues 32 + 2 x (active column tiles)
//     requests instead of 64;
//   * the MFMAs are issued from inline asm in a fixed order with the requests of the next chunk dealt out between them (hipcc keeps the
//     relative order of volatile asm statements): an LDS-DMA request costs its wave ~60 cycles of issue, which one MFMA's 8 free issue
//     cycles do not hide, but seven requests spread over 190 MFMAs do.
//
// STATUS: opt-in (GDRF_FWDT_W1=1), NOT the default.  Measured at the headline size: 8.7 ms against the 16-wave form's 8.2 ms.  With one wave
// per SIMD nobody covers a request's issue cost (~80 cycles each, 16 per 192 MFMAs - for A_k, where the same structure gained 40 %, it is 15
// per 480) nor the HBM latency of the W stream behind a one-chunk lookahead (two 64 KB buffers fill the LDS).  What it established: the
// immediate offset of global_load_lds is added to the LDS address as well as to the global one (four requests share one M0 write below;
// parity tests green), and that sharing M0 does not make a request cheaper (8.71 vs 8.75 ms).
//
// Hazards hipcc's tables do not see (inline asm): an accumulator is touched again 4 MFMAs later (same opcode, same destination: the
// hardware interlocks that case); the epilogue waits two s_nop 15 before it reads the accumulators.
#pragma once
#include <utility>
#include "gemm_split.h"
#include "gemm_tn_topics1.h"          // static_for, glds16_asm_s

namespace gdrf {

constexpr int FT1_A_BYTES = 2 * 2 * GDRF_TILE * 32 * 2;     // [2 row tiles][2 pieces][128][32] halfwords: 32 KB
constexpr int FT1_B_BYTES = 2 * 256 * 32 * 2;               // [2 pieces][256 columns][32] halfwords: 32 KB
constexpr int FT1_BUF = FT1_A_BYTES + FT1_B_BYTES;
constexpr int ft1_lds_bytes() { return 2 * FT1_BUF + 256 * 4; }

// LDS-DMA whose LDS destination is M0 + OFF (the instruction's immediate offset is added to the global address and to the LDS address alike:
// the caller passes a base that is OFF bytes low).  SET = false reuses the M0 of the request before: a write to M0 directly behind a DMA
// waits until that DMA has consumed it (~100 - 200 cycles, glds16_asm), so back-to-back requests with their own M0 each serialise.
template <int OFF, bool SET>
__device__ __forceinline__ void glds16_off(const void* sbase_minus_off, unsigned voff, unsigned m0_value) {
  if (SET) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(sbase_minus_off), "s"(m0_value), "n"(OFF) : "memory");
  else asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase_minus_off), "n"(OFF) : "memory");
}

// Grid and block map: those of fwd_t_split_q4_kernel (8 x K x rt8 workgroups; a workgroup = 256 rows of one topic).  Mp % 256 == 0.
__global__ __launch_bounds__(256, 1) void fwd_t_w1_kernel(FwdTSplitArgs<SplitF16> g) {
  using SP = SplitF16;
  using E = _Float16;
  using V8 = f16x8;
  constexpr int PIECE = GDRF_TILE * 32;                      // halfwords per 128-row piece image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int Mp = g.Mp;
  const unsigned xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned per_group = (unsigned)g.KG * (unsigned)g.rt8;
  const int grp = (int)(idx / per_group);
  const unsigned rem = idx - (unsigned)grp * per_group;
  const int kg = min(g.KG, g.K - grp * g.KG);
  const int64_t rt0 = 2 * ((int64_t)(rem / (unsigned)kg) * 8 + xcd);
  const int bz = grp * g.KG + (int)(rem % (unsigned)kg);
  const int64_t m0 = rt0 * GDRF_TILE;                        // 256 rows from here; a workgroup past the end runs on row 0 and stores nothing
  const bool interior = m0 + 256 <= g.nrows;

  const unsigned lds0 = lds_addr(smem);
  const int drow = lane >> 2, dq = ((lane & 3) ^ split_swz(drow)) * 8;
  const unsigned voff_a = (unsigned)((drow * Mp + dq) * 2), voff_b = (unsigned)((drow * 32 + dq) * 2);
  // requests of the chunk at reduction index kA of column pair cp into buffer `buf`.  A: 32 requests (row tile gi, piece p, 16-row block
  // rbk), 8 per wave: slot s -> request 8 w + s.  B: piece p, 16-column block blk < nb (the column tiles the chunk multiplies); wave w
  // takes the blocks blk = w, w + 4, ...: slot 8 + 2 i + p -> (p, blk = w + 4 i).
  auto dma_slot = [&](int cp, int kA, int nb, int buf, int slot) __attribute__((always_inline)) {
    const unsigned base = lds0 + (unsigned)(buf * FT1_BUF);
    if (slot < 8) {
      const int rq = 8 * w + slot, gi = rq >> 4, p = (rq >> 3) & 1, rbk = rq & 7;
      const int64_t row0 = m0 + gi * GDRF_TILE + rbk * 16;
      if (interior) {
        // slots 0 .. 7 of a wave are the 8 row blocks of ONE piece image (1 KB apart): two M0 values, four immediate offsets
        const char* src = reinterpret_cast<const char*>(g.Wh + p * g.w_stride + row0 * Mp + kA);
        const unsigned m0v = base + (unsigned)(((gi * 2 + p) * PIECE + (rbk & 4) * 512) * 2);
        switch (slot & 3) {
          case 0: if ((slot & 3) == 0) glds16_off<0, true>(src, voff_a, m0v); break;
          case 1: glds16_off<1024, false>(src - 1024, voff_a, m0v); break;
          case 2: glds16_off<2048, false>(src - 2048, voff_a, m0v); break;
          default: glds16_off<3072, false>(src - 3072, voff_a, m0v); break;
        }
      } else {
        int64_t row = row0 + drow;
        row = row < g.nrows ? row : 0;
        glds16_asm(g.Wh + p * g.w_stride + row * Mp + kA + dq, base + (unsigned)(((gi * 2 + p) * PIECE + rbk * 512) * 2));
      }
    } else {
      const int i = (slot - 8) >> 1, p = (slot - 8) & 1, blk = w + 4 * i;
      if (blk < nb)
        glds16_asm_s(g.STh + p * g.piece_stride + (((int64_t)bz * (Mp >> 5) + (kA >> 5)) * Mp + cp * 256 + blk * 16) * 32, voff_b,
                     base + (unsigned)(FT1_A_BYTES + (p * 256 * 32 + blk * 512) * 2));
    }
  };
  // column tiles (16 columns) of pair cp that the chunk at kA multiplies: j0 = cp 256 + 16 b <= kA + 31
  auto tiles_of = [&](int cp, int kA) { const int t = (kA + 31 - cp * 256) / 16 + 1; return t < 16 ? t : 16; };

  f32x4 acc[4][16];
  float rs[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) rs[a][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < 16; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  }
  auto mfma = [&](f32x4& c, const V8& a, const V8& b) __attribute__((always_inline)) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); };
  const int frag = lr * 32 + ((lg ^ split_swz(lr)) << 3);
  const int ncp = Mp / 256;

  {
    const int nb0 = tiles_of(0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) dma_slot(0, 0, nb0, 0, s);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  gdrf_raw_barrier();
  int buf = 0;
  // one chunk: NB = the number of column tiles it multiplies, a compile-time constant (a branch around the asm MFMAs makes hipcc copy the
  // accumulators - 16 v_accvgpr moves per skipped or taken tile): the first eight chunks of a column pair are the ramp NB = 2, 4, .., 16,
  // every later one multiplies all 16
  auto chunk = [&](auto nbc, int cp, int kA) __attribute__((always_inline)) {
    constexpr int NB = decltype(nbc)::value;
    int cp1 = cp, kA1 = kA + 32;                 // the chunk after this one (the last one requests nothing)
    if (kA1 >= Mp) { cp1 = cp + 1; kA1 = cp1 * 256; }
    const bool more = cp1 < ncp;
    const int nb1 = more ? tiles_of(cp1, kA1) : 0;
    const E* Ab = reinterpret_cast<const E*>(smem + buf * FT1_BUF) + (w >> 1) * 2 * PIECE;   // this wave's row tile (w / 2), rows 64 (w & 1) ..
    const E* Bb = reinterpret_cast<const E*>(smem + buf * FT1_BUF + FT1_A_BYTES);
    V8 fa[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[p][a] = *reinterpret_cast<const V8*>(Ab + p * PIECE + ((w & 1) * 64 + a * 16) * 32 + frag);
    // the 16 requests of the next chunk (8 of A, up to 8 of B) go out behind the MFMAs of the chunk's FIRST column tiles - two per tile
    // over the first eight of a full chunk, all of them behind the first tiles of a ramp chunk - so that they have the rest of the chunk
    // to land before its closing vmcnt(0) (issued behind the last tiles, their whole latency showed: bwd_wbar_f16_k64_kernel's lesson)
    constexpr int NQ = NB >= 16 ? 8 : (NB >= 4 ? NB / 2 : 1), PER = (16 + NQ - 1) / NQ;
    static_for<NB>([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      V8 fb[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) fb[p] = *reinterpret_cast<const V8*>(Bb + p * 256 * 32 + (b * 16) * 32 + frag);
#pragma unroll
      for (int x = 0; x < SP::NPROD; ++x)
#pragma unroll
        for (int a = 0; a < 4; ++a) mfma(acc[a][b], fa[SP::pa(x)][a], fb[SP::pb(x)]);
      if (more && b < NQ) {
#pragma unroll
        for (int s = b * PER; s < (b + 1 == NQ ? 16 : (b + 1) * PER); ++s) dma_slot(cp1, kA1, nb1, buf ^ 1, s);
      }
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    gdrf_raw_barrier();
    buf ^= 1;
  };
#pragma unroll 1
  for (int cp = 0; cp < ncp; ++cp) {
    static_for<8>([&](auto qc) __attribute__((always_inline)) { chunk(std::integral_constant<int, 2 * decltype(qc)::value + 2>{}, cp, cp * 256 + 32 * decltype(qc)::value); });
#pragma unroll 1
    for (int kA = cp * 256 + 256; kA < Mp; kA += 32) chunk(std::integral_constant<int, 16>{}, cp, kA);
    // fold the finished 256 columns into the row sums of squares, restart the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int b = 0; b < 16; b += 2) { s0 += acc[a][b][r] * acc[a][b][r]; s1 += acc[a][b + 1][r] * acc[a][b + 1][r]; }
        rs[a][r] += s0 + s1;
      }
#pragma unroll
      for (int b = 0; b < 16; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    }
  }
  float* rsum = reinterpret_cast<float*>(smem + 2 * FT1_BUF);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = group16_sum(rs[a][r]);
      if (lr == 0) rsum[w * 64 + a * 16 + lg * 4 + r] = v;
    }
  __syncthreads();
  {
    const int64_t m = m0 + tid;
    const SplitLay SL{g.K};
    const float un = g.sc[SL.w() + 1] * g.sc[SL.st(bz) + 1];
    if (m < g.nrows) g.tt[(int64_t)bz * g.ldt + m] = rsum[tid] * (un * un);
  }
}

}  // namespace gdrf
