// Batched kNN over an fp32 corpus on the matrix cores: 5..32 queries per corpus pass at (close to)
// the HBM rate, gfx950 (MI355X).
//
// Same contract as the scan kernels — steps 1-3 of ExactIndex.search (reference
// src/dewi/backends.py:420-444) on the reference's native dtype (fp32 rows, backends.py:403) — for
// query batches: `search_batch` on an fp32 corpus was vector-ALU-bound (8 queries per pass 0.68 ms at
// 1M x 768); on the matrix cores 32 queries cost one corpus pass.
//
// ARITHMETIC (DEWI_F32_SPLIT, dims up to 1024).  gfx950 has no fast fp32 matrix instruction:
// v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (157 TFLOP/s), and 32 queries x 1M x 768 are
// 49 GFLOP = 0.31 ms of a saturated matrix pipe beside a 0.38 ms HBM pass — the first version of this
// kernel sat in that corner (pipe busy 70 %, clock down to 1.7 GHz, pass 0.50-0.53 ms).  Now every fp32
// value is cut into three bf16 pieces, x = hi + mid + lo EXACTLY (8 + 8 + 8 significand bits), and a
// block of products is six v_mfma_f32_32x32x16_bf16 (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi,
// smallest first) with exact products and fp32 accumulation.  The three dropped terms (mid*lo, lo*mid,
// lo*lo) are below 2^-23 |q_i e_i| each — one fp32 rounding of that product — so scores stay fp32-grade
// (tests/test_hip_mfma_f32.py bounds |score - f64| <= 3e-7 on heavy-tailed vectors; NaN stays NaN).
// Per chunk and wave: 12 matrix instructions of 32 cycles instead of 16 of 64, plus 72 vector
// instructions for the cut (v_cvt_pk_bf16_f32 / shift / and / v_pk_add_f32): the pass drops to
// 0.445-0.48 ms (6.4-6.9 TB/s).  Dim 1536 would need 144 registers of query pieces per lane and keeps
// v_mfma_f32_32x32x2_f32 on the fp32 values (exact products, an fmaf chain per depth slice).
//
// SPACE l2 (reference backends.py:434-436, -sum((E - q)^2)): scored as 2<e,q> - ||e||^2 - ||q||^2.  <e,q> is the
// same matrix product on the raw rows and raw (bf16 corpus: bf16-rounded) queries; ||q||^2 comes float64-summed
// from the query preparation kernel; ||e||^2 is summed in this kernel from the very fragments it multiplies (16
// fmas per chunk and lane, shares of the 8 waves x 2 lane halves exchanged through the spare slot of the reduction
// buffer and summed in a fixed order), so the corpus is still read once and nothing is stored per row.  The three
// terms round at the magnitude of ||e||^2 + ||q||^2: an ABSOLUTE error of about an ulp of that magnitude, where the
// reference's fp32 sum of squared differences has a relative error of the distance itself.  For rows far from the query
// the two agree to fp32 noise; a near-duplicate of the query scores +-1e-4 here (||e||^2 ~ 500) and ~0 there.  So:
//   * fp32 corpus: EXACT-REFINE mode (default).  |score - exact| <= l2_margin (||e||^2 + ||q||^2), l2_margin = 2 dim 2^-23
//     (dim fp32 accumulations of products bounded by (e_i^2 + q_i^2) / 2, twice, the norm's own sum, three roundings).  The
//     sample pass records score - margin, the filter passes score + margin >= threshold, and the select kernel
//     (select_rerank.hip, refine_top_candidates) widens the cut by the bound and re-scores the candidates with the row
//     kernels' arithmetic: the batch equals the one-query searches bit for bit (tests/test_hip_round3.py).
//   * bf16 corpus: the form above unrefined is an OPT-IN (dewi_tuning_set batched_mfma = 2); by default l2 batches over a
//     bf16 corpus take the exact row-per-wave kernels (abi.cpp plan_batch).  tests/test_hip_mfma_f32.py opts in and
//     compares with the oracle at gaps and tolerances scaled by that magnitude.  fp32
// corpora: dims up to 768 (beyond, query pieces + norm traffic do not fit the registers: such batches keep the
// row-per-wave l2 kernels); bf16 corpora: every supported dim.  1 M x 768: fp32 32 queries 0.49 ms per pass
// (cosine 0.47), bf16 0.225 ms — l2 batches used to cost a row-kernel pass per 4 (fp32) queries.
//
// Roofline: HBM.  Algorithmic bytes per pass = n_rows * dim * 4 (the corpus read once), 0.38 ms at
// 8 TB/s for 1M x 768; flops = 6 * 2 * 32 * n_rows * dim bf16 (0.29 PFLOP) = 0.12 ms at 2.5 PFLOP/s.
//
// Structure (one 8-wave workgroup per CU, persistent over 32-document tiles):
//  * A tile is cut into CHUNKS of 32 rows x 256 columns (32 KiB): 32 DMA pieces of 1 KiB = one row's
//    256 columns each (buffer_load_dwordx4 ... lds, global -> LDS without registers), ring of four
//    chunks: one being multiplied, three in flight (96 KiB per CU).
//  * THE EIGHT WAVES SPLIT THE DEPTH: wave w multiplies columns [32w, 32w+32) of every chunk, so its
//    share of the 32 normalised queries is dim/8 columns (as three bf16 pieces: 72 registers per lane
//    at dim 768) and lives in registers for the whole kernel.  Per chunk and wave: 4 ds_read_b128 (lane
//    (r, h) takes columns 32w + 8m + 4h .. +3 of row r, m = 0..3); reads 2p and 2p+1 make the eight
//    k-slots of lane half h in MFMA group p (the k index of an MFMA is only a pairing of A and B lanes,
//    so any column may stand at any k as long as both sides agree).
//  * BANK CONFLICTS: chunk rows are 1 KiB apart, so the 16 lanes of a ds_read_b128 group (16 rows, same
//    column unit) would all hit the same banks.  The LDS image is linear per piece and the SOURCE
//    address is permuted: 16-byte unit u of row r is stored at unit u ^ (r & 15); reads apply the same
//    XOR -> conflict-free.
//  * At the end of a tile the eight partial 32 x 32 blocks are summed through LDS in a fixed order
//    (wave 0 .. 7: bit-reproducible): wave w receives accumulator registers 2w and 2w+1 — two
//    documents per lane — and filters them against the query's threshold.
//  * THRESHOLDS AND SURVIVORS as in the bf16 matrix-core path (knn_mfma_bf16.hip): a sample pass of the
//    same kernel over every `sample_stride`-th tile keeps group maxima, their c-th largest is a valid
//    lower bound of the query's c-th best score; the full pass stores every score that is not below it
//    as a raw record (row << 32 | score bits) into the (workgroup, query) segment — a survivor waits
//    in its lane until a second one needs the same place, then the wave's waiting records go out
//    together, slots from a per-query counter in LDS — and the select kernel finishes exactly.  A segment that overflows (adversarial corpora) marks the query
//    (count > capacity): ids -1, the caller re-runs it on the exact small-batch kernels.
#include "select_common.hpp"

namespace dewi {

typedef float f32x16f __attribute__((ext_vector_type(16)));
typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));

constexpr int kF32Threads = 512;
constexpr int kF32Waves = kF32Threads / kWave;       // 8: the depth split
constexpr int kF32TileRows = 32;
constexpr int kF32Queries = 32;                      // queries per pass (one MFMA column block)
constexpr int kF32ChunkCols = 256;
// Geometry of the two element types this kernel is built for.  fp32: a chunk row is 1 KiB = one DMA piece, chunk
// 32 KiB, ring of 4 (3 chunks = 96 KiB in flight), per wave and chunk 4 pieces, 4 ds_read_b128 and 16 MFMAs
// (32x32x2 f32).  bf16 (small query batches over a bf16 corpus; also dim 1024 / 1536): a chunk row is 512 B, a DMA
// piece two rows, chunk 16 KiB, ring of 8 (7 chunks = 112 KiB in flight), per wave and chunk 2 pieces, 2
// ds_read_b128 and 2 MFMAs (32x32x16 bf16) — a pure tile-delivery kernel: 49 GFLOP of bf16 are 20 us of matrix pipe.
template <bool BF16>
struct DepthGeo {
  static constexpr int kElem = BF16 ? 2 : 4;
  static constexpr int kRowChunk = kF32ChunkCols * kElem;             // bytes of a row inside a chunk: 1024 / 512
  static constexpr int kChunk = kF32TileRows * kRowChunk;             // 32 KiB / 16 KiB
  static constexpr int kRing = BF16 ? 8 : 4;
  static constexpr int kPieces = kChunk / 1024 / kF32Waves;           // 1 KiB DMA pieces per wave and chunk: 4 / 2
  static constexpr int kReads = BF16 ? 2 : 4;                         // ds_read_b128 per wave and chunk (== kPieces)
  static constexpr int kWaitPieces = (kRing - 2) * kPieces;           // vmcnt that leaves chunk g+1 landed: 8 / 12
};
constexpr int kF32MaxL2Chunks = 3;                                  // fp32 corpus, l2 space: dims up to 768 (CH = 4 spills 12 bytes per lane)
constexpr int kF32RedRegs = 15;                                     // LDS slots per wave: the 14 registers it hands over + a spare
constexpr int kF32RedBytes = kF32Waves * kF32RedRegs * kWave * 4;   // 30 KiB
template <bool BF16>
constexpr int depth_lds_bytes() { return DepthGeo<BF16>::kRing * DepthGeo<BF16>::kChunk + kF32RedBytes + kF32Queries * 4; }
// fp32 corpus arithmetic (see ARITHMETIC above): 1 = three-piece bf16 cut + six bf16 MFMAs, 0 = v_mfma_f32_32x32x2_f32.
#ifndef DEWI_F32_SPLIT
#define DEWI_F32_SPLIT 1
#endif
#ifndef DEWI_F32_PENDING2
#define DEWI_F32_PENDING2 0   // 1 = two waiting places per accumulator register instead of one (a wave then flushes when a THIRD
                              // survivor meets two waiting ones: fewer flushes).  Round 4, three interleaved rounds on one box
                              // (scripts/probes/r04_f32_pending.sh, 1 M x 768, 32 queries): pass 0.4657-0.4667 ms with two places
                              // against 0.4600-0.4606 with one — SLOWER by 6 us: the flush count is not what the pass waits for;
                              // the select chains that keep two places cost more than the flushes they save.  Off.
#endif
#ifndef DEWI_F32_DEFER_FILTER
#define DEWI_F32_DEFER_FILTER 1   // filter (and survivor stores) of a tile one chunk later, right behind a barrier
#endif
#ifndef DEWI_F32_ABLATE_STORES
#define DEWI_F32_ABLATE_STORES 0   // timing experiments only: 1 = survivors take their slots but are not stored (wrong results)
#endif
#ifndef DEWI_F32MFMA_DMA_AUX
#define DEWI_F32MFMA_DMA_AUX 2   // non-temporal tile DMA: the corpus is read once
#endif

using LdsPtrF = void __attribute__((address_space(3)))*;

typedef float f32x2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16_rne(f32x2f v) {      // v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2f));
}
__device__ __forceinline__ f32x2f widen_bf16(uint32_t pk) {
  return f32x2f{__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
}
// Eight fp32 values (two 16-byte reads) -> three bf16x8 operands with x = hi + mid + lo exactly: x - hi is exact
// (hi holds x's leading 8 bits, rounded), at most 16 bits long; mid takes its leading 8, what is left fits the 8
// bits of lo.  9 vector instructions per pair of values (cvt_pk, shift, and, pk_add; twice; cvt_pk).  NaN stays NaN.
__device__ __forceinline__ void split3(const u32x4f& x0, const u32x4f& x1, u32x4f& hi, u32x4f& mid, u32x4f& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t ua = j < 2 ? x0[2 * j] : x1[2 * j - 4], ub = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
    const f32x2f x = {__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub)};
    hi[j] = pack_bf16_rne(x);
    const f32x2f r1 = x - widen_bf16(hi[j]);
    mid[j] = pack_bf16_rne(r1);
    const f32x2f r2 = r1 - widen_bf16(mid[j]);
    lo[j] = pack_bf16_rne(r2);
  }
}

// CH = dim / 256 chunks per tile.  Tiles of this launch: t = (blockIdx.x + i * gridDim.x) * tile_stride.
// SAMPLE: out is a float array [32][out_stride]: out[q * out_stride + (16 * blockIdx.x + 2 * wave + h) * 2 + e] =
//         the best score this lane's register e saw over the workgroup's tiles (32 group maxima per workgroup).
// filter: raw records out[(blockIdx.x * 32 + q) * out_stride + slot], cnt[blockIdx.x * 32 + q] = records offered.
// BF16: E and Qn are bf16 (Qn prepared by prepare_queries_bf16); otherwise fp32.
// L2: the score is -||e - q||^2 (reference backends.py:434-436) in the form 2<e,q> - ||e||^2 - ||q||^2: Qn holds the raw
//     (bf16 corpus: bf16-rounded) queries, qn2 their squared norms, and ||e||^2 is summed here from the very fragments
//     that are multiplied (see "row norms" below).
// PARTIAL (round 4): the row's LAST chunk holds only last_cols of its 256 columns (dim = 256 (CH - 1) + last_cols; whole 16-byte
//     units: a multiple of 4 columns of fp32, 8 of bf16), so that every such dim takes this pass, not only whole chunks.  Wave w
//     multiplies columns [32 w, 32 w + 32) of a chunk: in the last chunk the waves with 32 w >= last_cols have nothing of this row —
//     they still keep every barrier, but skip their matrix instructions — and the ONE wave whose slice the row ends in clears
//     the fragment units behind the row's end (DMA lanes behind the end move nothing, so those LDS bytes are whatever an
//     earlier chunk left there): a NaN there must not reach this row's score through a 0 x NaN product.  Row stride and query
//     stride become run-time values.  Cosine, and l2 over an fp32 corpus (exact-refine mode: the select re-scores with
//     scan_rows_any's arithmetic, widths from 132 columns = 33 units).
template <bool BF16, int CH, bool SAMPLE, bool L2, bool PARTIAL = false>
__global__ __launch_bounds__(kF32Threads, 2) void mfma_scan_f32(const void* __restrict__ E, int64_t n_rows,
                                                                const void* __restrict__ Qn, int64_t n_tiles,
                                                                int64_t tile_stride, const float* __restrict__ thr,
                                                                uint64_t* __restrict__ out, int64_t out_stride,
                                                                uint32_t* __restrict__ cnt, int n_active,
                                                                const float* __restrict__ qn2, float aux, int last_cols) {
  // aux: l2 — the error bound per unit of ||e||^2 + ||q||^2 (exact-refine mode, 0 = unrefined); cosine — a bias subtracted
  // from every threshold (0, or two error bounds when the pass pre-selects over the bf16 shadow of an fp32 corpus)
#if defined(__HIP_DEVICE_COMPILE__)
  const float l2_margin = L2 ? aux : 0.f;
  using G = DepthGeo<BF16>;
  static_assert(!(PARTIAL && L2 && BF16), "partial last chunk + l2: fp32 corpora (exact-refine mode) only");
  const int DIM = PARTIAL ? (CH - 1) * kF32ChunkCols + last_cols : CH * kF32ChunkCols;   // compile-time unless PARTIAL
  constexpr int kUnitCols = BF16 ? 8 : 4;                      // columns per 16-byte unit
  constexpr int RM = G::kRing - 1;                             // ring slot of chunk g: g & RM
  extern __shared__ __attribute__((aligned(16))) char lds[];   // ring | partial sums | per-query counters
  float* const red = reinterpret_cast<float*>(lds + G::kRing * G::kChunk);
  uint32_t* const lcnt = reinterpret_cast<uint32_t*>(lds + G::kRing * G::kChunk + kF32RedBytes);

  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- this wave's share of the queries (B operand), 16 bytes per read m: lane (r, h) holds
  //      fp32: Qn[r][256 ch + 32 w + 8 m + 4 h .. +3]  (4 MFMAs)     bf16: Qn[r][256 ch + 32 w + 16 m + 8 h .. +7]  (1 MFMA)
  //      split fp32: reads 2p and 2p+1 are cut into qf[ch][3p + 0 / 1 / 2] = hi / mid / lo (8 columns per operand)
  constexpr bool kSplit = !BF16 && DEWI_F32_SPLIT && CH <= 4;   // dim 1536: three query pieces would not fit the registers
  constexpr int kQRegs = kSplit ? 6 : G::kReads;
  u32x4f qf[CH][kQRegs];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    u32x4f raw[G::kReads];
#pragma unroll
    for (int m = 0; m < G::kReads; ++m) {
      const char* qrow = static_cast<const char*>(Qn) + static_cast<int64_t>(r) * DIM * G::kElem;
      raw[m] = u32x4f{0u, 0u, 0u, 0u};
      // (PARTIAL, last chunk: units behind the row's end stay zero — whole waves, and the tail of the wave the row ends in)
      const bool behind = PARTIAL && ch == CH - 1 && 32 * w + kUnitCols * (2 * m + h) >= last_cols;
      if (!behind) raw[m] = *reinterpret_cast<const u32x4f*>(qrow + kF32ChunkCols * G::kElem * ch + 32 * G::kElem * w + 16 * (2 * m + h));
    }
    if constexpr (kSplit) {
      split3(raw[0], raw[1], qf[ch][0], qf[ch][1], qf[ch][2]);
      split3(raw[2], raw[3], qf[ch][3], qf[ch][4], qf[ch][5]);
    } else {
#pragma unroll
      for (int m = 0; m < G::kReads; ++m) qf[ch][m] = raw[m];
    }
  }
  const float thr_l = SAMPLE ? -__builtin_inff() : (r < n_active ? thr[r] - (L2 ? 0.f : aux) : __builtin_inff());
  const float qn2_l = L2 ? qn2[r] : 0.f;
  // pin the waits for these loads here, before any DMA is in flight (a compiler-inserted vmcnt(0) inside the
  // chunk loop would drain the ring on every iteration)
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
#pragma unroll
    for (int m = 0; m < kQRegs; ++m) asm volatile("" ::"v"(qf[ch][m]));
  }
  asm volatile("" ::"v"(thr_l), "v"(qn2_l));
  if (!SAMPLE && threadIdx.x < kF32Queries) lcnt[threadIdx.x] = 0;   // read first at a tile end, behind several barriers

  // ---- DMA.  The 16-byte unit u of a chunk row lands at LDS unit u ^ (row & 15), i.e. the lane that fills LDS
  // unit u of that row fetches unit u ^ (row & 15).
  //   fp32: piece p (0..3) of a wave is row w + 8 p (one row = 64 units); row & 15 is w for p even, w + 8 for p odd.
  //   bf16: piece p (0..1) is rows 2 (w + 8 p) and + 1 (32 units each; lanes 32.. take the second row);
  //         row & 15 = (2 w + lane / 32) & 15 for both pieces.
  const uint32_t row_bytes = static_cast<uint32_t>(DIM) * G::kElem;
  const uint32_t row_b = static_cast<uint32_t>(2 * w + (lane >> 5));
  const uint32_t voff_even = BF16 ? row_b * row_bytes + 16u * (static_cast<uint32_t>(lane & 31) ^ (row_b & 15u))
                                  : static_cast<uint32_t>(w) * row_bytes + 16u * static_cast<uint32_t>(lane ^ w);
  const uint32_t voff_odd = BF16 ? voff_even
                                 : static_cast<uint32_t>(w) * row_bytes + 16u * static_cast<uint32_t>(lane ^ (w + 8));
  const char* Eb = reinterpret_cast<const char*>(E);
  const int64_t first = static_cast<int64_t>(blockIdx.x), step = static_cast<int64_t>(gridDim.x);
  const int64_t n_my = first < n_tiles ? (n_tiles - first + step - 1) / step : 0;
  // descriptor of this workgroup's it-th tile: base = the tile's first row, size = its valid bytes, so rows past
  // the end of the corpus — and whole tiles past this workgroup's last one (size 0) — read as zeros
  auto tile_rsrc = [&](int64_t it) {
    int64_t row0 = 0;
    int valid = 0;
    if (it < n_my) {
      row0 = (first + it * step) * tile_stride * kF32TileRows;
      const int64_t left = n_rows - row0;
      valid = left < kF32TileRows ? static_cast<int>(left) : kF32TileRows;
    }
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Eb + row0 * row_bytes), 0, valid * static_cast<int>(row_bytes),
                                             0x00020000);
  };
  // ---- A-fragment read addresses inside ring slot 0: row r, unit (units-per-wave * w + 2 m + h) ^ (r & 15)
  uint32_t a_addr[G::kReads];
#pragma unroll
  for (int m = 0; m < G::kReads; ++m)
    a_addr[m] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsPtrF)(lds))) +
                static_cast<uint32_t>(r * G::kRowChunk + 16 * (((G::kRowChunk / 128) * w + 2 * m + h) ^ (r & 15)));

  // ---- Software pipeline.  Chunk sequence of this workgroup: g = it * CH + ch -> ring slot g & 3.  Iteration g
  //   waits for chunk g+1 (own pieces: vmcnt, everybody's: barrier), starts the LDS reads of chunk g+1, and
  //   MULTIPLIES CHUNK g FROM REGISTERS (read during iteration g-1), so the matrix pipe starts right behind the
  //   barrier; the four DMA pieces of chunk g+4 go out BETWEEN the MFMAs — an fp32 MFMA occupies the pipe for 64
  //   cycles (128 with the partner wave's in between), which covers a piece's issue — instead of in front of them,
  //   where both waves of a SIMD left the pipe idle together (first version of this kernel: matrix pipe busy 60 %).
  //   Chunk g+4 refills the slot of chunk g, which every wave has finished reading before the barrier of
  //   iteration g.  Chunks past the last tile are fetched through an empty descriptor (zeros, no memory traffic)
  //   so that every iteration has the same 12 pieces outstanding at its vmcnt(8).
  // PARTIAL: in the row's last chunk only the lanes whose SOURCE unit lies inside the row fetch (EXEC-masked DMA: the other
  // lanes move nothing and leave their LDS bytes as they are — only the waves that skip the chunk would read those).  Without the
  // mask a piece carried the head of the next row behind the row's tail: dim 128 moved every byte twice (pass 0.91 ms per 3 GB).
  const uint32_t valid_units = PARTIAL ? static_cast<uint32_t>(last_cols / kUnitCols) : 64u;
  const bool in_row_even = BF16 ? ((static_cast<uint32_t>(lane & 31) ^ (row_b & 15u)) < valid_units)
                                : (static_cast<uint32_t>(lane ^ w) < valid_units);
  const bool in_row_odd = BF16 ? in_row_even : (static_cast<uint32_t>(lane ^ (w + 8)) < valid_units);
  auto issue_piece = [&](__amdgpu_buffer_rsrc_t rsrc, int ch, int slot, int p) {
    char* base = lds + slot * G::kChunk + w * 1024;        // piece w + 8 p of the chunk's 1 KiB pieces
    if (PARTIAL && ch == CH - 1) {
      if ((p & 1) ? in_row_odd : in_row_even)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtrF)(base + p * 8 * 1024), 16, (p & 1) ? voff_odd : voff_even,
                                                 p * (BF16 ? 16 : 8) * static_cast<int>(row_bytes) + ch * G::kRowChunk, 0,
                                                 DEWI_F32MFMA_DMA_AUX);
      return;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtrF)(base + p * 8 * 1024), 16, (p & 1) ? voff_odd : voff_even,
                                             p * (BF16 ? 16 : 8) * static_cast<int>(row_bytes) + ch * G::kRowChunk, 0,
                                             DEWI_F32MFMA_DMA_AUX);
  };
  auto issue_g = [&](int64_t g) {   // run-time chunk index (prologue only)
    const int64_t it = g / CH;
    const int ch = static_cast<int>(g - it * CH);
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(it);
    const int slot = static_cast<int>(g & RM);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (ch == c) {
#pragma unroll
        for (int p = 0; p < G::kPieces; ++p) issue_piece(rs, c, slot, p);
      }
    }
  };
  auto read_chunk = [&](u32x4f (&dst)[G::kReads], int slot) {
    const uint32_t slot_off = static_cast<uint32_t>(slot) * G::kChunk;
#pragma unroll
    for (int m = 0; m < G::kReads; ++m) asm volatile("ds_read_b128 %0, %1" : "=v"(dst[m]) : "v"(a_addr[m] + slot_off));
  };
  // partial sums: wave v keeps register j at red[(v * 15 + slot_of(j, v)) * 64 + lane]; its own two registers go
  // to the spare slot 14 (never read), which keeps the stores free of branches
  const uint32_t red_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsPtrF)(red))) + 4u * static_cast<uint32_t>(lane);
  float own0 = 0.f, own1 = 0.f;
  // ROW NORMS (L2).  Lane (r, h) sees 16 of row r's 256 columns per chunk; their squares are summed into nrm_a / nrm_b (
  // fmas, fixed order) over the tile's chunks, and the lane's share of ||e_r||^2 goes to its place in the wave's spare
  // slot 14, BEHIND the two placeholder stores of the loop below (LDS operations of a wave execute in order).
  // Stage 2 sums the sixteen shares of a row (eight waves x two lane halves) in a fixed order.
  // (scalar fmas on two accumulators: __builtin_elementwise_fma on a float2 came out of hipcc 7.2 as v_pk_fma_f32 pairs
  // with op_sel_hi:[0,0,1] that add the LOW element's square to both halves — wrong sums)
  float nrm_a = 0.f, nrm_b = 0.f;
  auto add_squares = [&](const u32x4f& x) {
#pragma unroll
    for (int i = 0; i < (BF16 ? 4 : 2); ++i) {
      float va, vb;
      if constexpr (BF16) {
        va = __uint_as_float(x[i] << 16);
        vb = __uint_as_float(x[i] & 0xffff0000u);
      } else {
        va = __uint_as_float(x[2 * i]);
        vb = __uint_as_float(x[2 * i + 1]);
      }
      nrm_a = __builtin_fmaf(va, va, nrm_a);
      nrm_b = __builtin_fmaf(vb, vb, nrm_b);
    }
  };
  auto stage1_store = [&](const f32x16f& acc) {     // hand the other waves their registers of this wave's partial block
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int sl = j < 2 * w ? j : (j > 2 * w + 1 ? j - 2 : 14);
      asm volatile("ds_write_b32 %0, %1" ::"v"(red_addr + static_cast<uint32_t>((w * kF32RedRegs + sl) * kWave * 4)), "v"(acc[j]) : "memory");
    }
    if constexpr (L2) {
      const float rown = nrm_a + nrm_b;              // this lane's half (h) of the wave's share of ||e_r||^2
      asm volatile("ds_write_b32 %0, %1" ::"v"(red_addr + static_cast<uint32_t>((w * kF32RedRegs + 14) * kWave * 4)), "v"(rown) : "memory");
    }
#pragma unroll
    for (int ww = 0; ww < kF32Waves; ++ww) {
      if (w == ww) {
        own0 = acc[2 * ww];
        own1 = acc[2 * ww + 1];
      }
    }
  };
  float part[kF32Waves][2];
  // documents of registers 2w, 2w+1 inside the tile: 2 (w & 1) + 8 (w >> 1) + 4 h and the next row
  const uint32_t nrm_addr = red_addr - 4u * static_cast<uint32_t>(lane) +
                            4u * static_cast<uint32_t>(2 * (w & 1) + 8 * (w >> 1) + 4 * h);
  auto stage2_load = [&]() {                         // registers 2w, 2w+1 of every wave's partial block (own: spare slot)
#pragma unroll
    for (int v = 0; v < kF32Waves; ++v) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * w + e;
        const int sl = v == w ? 14 : (w < v ? j : j - 2);
        asm volatile("ds_read_b32 %0, %1" : "=v"(part[v][e]) : "v"(red_addr + static_cast<uint32_t>((v * kF32RedRegs + sl) * kWave * 4)));
      }
    }
  };
  auto stage2_pin = [&]() {                          // the loads above are complete (callers wait lgkmcnt(0) first)
#pragma unroll
    for (int v = 0; v < kF32Waves; ++v) {
      asm volatile("" : "+v"(part[v][0]), "+v"(part[v][1]));
    }
  };
  float mx0 = -__builtin_inff(), mx1 = -__builtin_inff();
  // survivors waiting in this lane: raw records (row << 32 | score bits) of query r go to the (workgroup, query)
  // segment, the slot comes from the query's counter in LDS
  constexpr uint32_t kNoDoc = 0xffffffffu;
  uint32_t pend_doc0 = kNoDoc, pend_doc1 = kNoDoc;
  float pend_s0 = 0.f, pend_s1 = 0.f;
#if DEWI_F32_PENDING2
  // a SECOND waiting place per accumulator register (round 4): the wave flushes when a THIRD survivor meets two waiting
  // ones, which takes more than twice as many survivors as two meeting (birthday scaling: ~14 -> ~40 per wave)
  uint32_t pend_doc0b = kNoDoc, pend_doc1b = kNoDoc;
  float pend_s0b = 0.f, pend_s1b = 0.f;
#endif
  auto flush_pending = [&]() {
    const uint32_t cap = static_cast<uint32_t>(out_stride);
    uint64_t* seg = out + (static_cast<int64_t>(blockIdx.x) * kF32Queries + r) * out_stride;
    // Inline asm: for a plain atomicAdd on LDS hipcc first waits for vmcnt(0) — every LDS-DMA piece in flight could
    // alias the counter as far as it knows — which would drain this wave's ring on every flush.
    const uint32_t cnt_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsPtrF)(lcnt))) + 4u * static_cast<uint32_t>(r);
    auto take_slot = [&]() {
      uint32_t slot;
      asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(cnt_addr), "v"(1u) : "memory");
      return slot;
    };
    if (pend_doc0 != kNoDoc) {
      const uint32_t slot = take_slot();
      if (slot < cap && !DEWI_F32_ABLATE_STORES) seg[slot] = (static_cast<uint64_t>(pend_doc0) << 32) | __float_as_uint(pend_s0);
      pend_doc0 = kNoDoc;
    }
    if (pend_doc1 != kNoDoc) {
      const uint32_t slot = take_slot();
      if (slot < cap && !DEWI_F32_ABLATE_STORES) seg[slot] = (static_cast<uint64_t>(pend_doc1) << 32) | __float_as_uint(pend_s1);
      pend_doc1 = kNoDoc;
    }
#if DEWI_F32_PENDING2
    if (pend_doc0b != kNoDoc) {
      const uint32_t slot = take_slot();
      if (slot < cap && !DEWI_F32_ABLATE_STORES) seg[slot] = (static_cast<uint64_t>(pend_doc0b) << 32) | __float_as_uint(pend_s0b);
      pend_doc0b = kNoDoc;
    }
    if (pend_doc1b != kNoDoc) {
      const uint32_t slot = take_slot();
      if (slot < cap && !DEWI_F32_ABLATE_STORES) seg[slot] = (static_cast<uint64_t>(pend_doc1b) << 32) | __float_as_uint(pend_s1b);
      pend_doc1b = kNoDoc;
    }
#endif
  };
  auto stage2_finish = [&](int64_t it) {             // fixed summation order 0..7 whichever wave sums; then the filter
    float s0 = 0.f, s1 = 0.f;
    float m0 = 0.f, m1 = 0.f;                        // l2: bound of |score - (-||e - q||^2)|, see l2_margin
#pragma unroll
    for (int v = 0; v < kF32Waves; ++v) {
      const float p0 = v == w ? own0 : part[v][0], p1 = v == w ? own1 : part[v][1];
      s0 = v == 0 ? p0 : s0 + p0;
      s1 = v == 0 ? p1 : s1 + p1;
    }
    if constexpr (L2) {                                // -||e - q||^2 = 2 <e,q> - ||e||^2 - ||q||^2, one rounding per step
      // the eight shares of ||e||^2 of this lane's two documents, read here (not with the partial sums at the top of
      // the chunk: sixteen more live registers through the MFMA block would spill at dim 1024) and summed in wave order
      float n0 = 0.f, n1 = 0.f;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {           // two waves' shares (two lane halves each) at a time: eight transient registers
        float np[2][2][2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
              asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(np[v][hh][e])      // one address register, immediate offsets
                           : "v"(nrm_addr), "n"(((2 * pair + v) * kF32RedRegs + 14) * kWave * 4 + 128 * hh + 4 * e));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            asm volatile("" : "+v"(np[v][hh][0]), "+v"(np[v][hh][1]));
            n0 = (pair == 0 && v == 0 && hh == 0) ? np[v][hh][0] : n0 + np[v][hh][0];
            n1 = (pair == 0 && v == 0 && hh == 0) ? np[v][hh][1] : n1 + np[v][hh][1];
          }
        }
      }
      s0 = (2.f * s0 - n0) - qn2_l;
      s1 = (2.f * s1 - n1) - qn2_l;
      // EXACT-REFINE mode (fp32 corpus: l2_margin = 2 dim 2^-23): the three terms above carry an absolute error of at most
      // l2_margin * (||e||^2 + ||q||^2) — dim fp32 accumulations of products bounded by (e_i^2 + q_i^2) / 2, twice, plus the
      // norm's own sum and three roundings.  The sample pass records score - margin (the c-th largest of those is a lower
      // bound of the c-th best EXACT score), the filter passes score + margin >= threshold (no row of the exact top c is
      // lost), and the select kernel re-scores the candidates exactly (select_rerank.hip, refine_top_candidates).
      m0 = l2_margin * (n0 + qn2_l);
      m1 = l2_margin * (n1 + qn2_l);
    }
    const int64_t row0 = (first + it * step) * tile_stride * kF32TileRows;
    const int64_t doc = row0 + 2 * (w & 1) + 8 * (w >> 1) + 4 * h;     // register 2w; register 2w+1 is the next row
    if (doc >= n_rows) s0 = -__builtin_inff();                         // padding rows of the last tile never pass
    if (doc + 1 >= n_rows) s1 = -__builtin_inff();
    if constexpr (SAMPLE) {
      mx0 = __builtin_fmaxf(mx0, s0 - m0);
      mx1 = __builtin_fmaxf(mx1, s1 - m1);
    } else {
      // A survivor waits in its lane (one place per accumulator register) until a second one arrives for the same
      // place somewhere in the wave; then the whole wave's waiting records go out together.  A vector store issued
      // into the full DMA queue costs the workgroup ~0.1 us at its next barrier whether it carries one record or
      // sixty-four (a batch of 32 ran 40 us behind a batch of 8 on per-survivor stores; taking the LDS slots
      // alone cost nothing), and a wave collects ~14 survivors before two meet.
      const bool pass0 = !(s0 + m0 < thr_l), pass1 = !(s1 + m1 < thr_l);   // NaN passes (NumPy ranks NaN first)
#if DEWI_F32_PENDING2
      if (__builtin_amdgcn_ballot_w64((pass0 && pend_doc0 != kNoDoc && pend_doc0b != kNoDoc) ||
                                      (pass1 && pend_doc1 != kNoDoc && pend_doc1b != kNoDoc)) != 0ull) flush_pending();
      if (pass0) {
        const bool first_free = pend_doc0 == kNoDoc;
        pend_doc0b = first_free ? pend_doc0b : static_cast<uint32_t>(doc);
        pend_s0b = first_free ? pend_s0b : s0;
        pend_doc0 = first_free ? static_cast<uint32_t>(doc) : pend_doc0;
        pend_s0 = first_free ? s0 : pend_s0;
      }
      if (pass1) {
        const bool first_free = pend_doc1 == kNoDoc;
        pend_doc1b = first_free ? pend_doc1b : static_cast<uint32_t>(doc + 1);
        pend_s1b = first_free ? pend_s1b : s1;
        pend_doc1 = first_free ? static_cast<uint32_t>(doc + 1) : pend_doc1;
        pend_s1 = first_free ? s1 : pend_s1;
      }
#else
      if (__builtin_amdgcn_ballot_w64((pass0 && pend_doc0 != kNoDoc) || (pass1 && pend_doc1 != kNoDoc)) != 0ull) flush_pending();
      if (pass0) {
        pend_doc0 = static_cast<uint32_t>(doc);
        pend_s0 = s0;
      }
      if (pass1) {
        pend_doc1 = static_cast<uint32_t>(doc + 1);
        pend_s1 = s1;
      }
#endif
    }
  };

  // l2 keeps the filter at the end of the iteration: it reads the row norms out of the reduction buffer, which the
  // other waves refill at the end of the tile's last chunk (two chunks per tile: no barrier in between); with ONE chunk
  // per tile (dim 256) there is no later chunk of the same tile to move it to (the wave's own two registers of the
  // partial block, own0 / own1, are replaced at the end of the next iteration)
  constexpr bool kDeferFilter = DEWI_F32_DEFER_FILTER && !L2 && CH >= 2;
  u32x4f cur[G::kReads], nxt[G::kReads];
  f32x16f acc;
  auto reads_done = [&](u32x4f (&f)[G::kReads]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < G::kReads; ++m) asm volatile("" : "+v"(f[m]));
  };
  if (n_my > 0) {
#pragma unroll
    for (int g0 = 0; g0 < G::kRing - 1; ++g0) issue_g(g0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::kWaitPieces) : "memory");        // own pieces of chunk 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_g(G::kRing - 1);
    read_chunk(cur, 0);
    reads_done(cur);
  }
  for (int64_t it = 0; it < n_my; ++it) {
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const int64_t g = it * CH + ch;
      // own pieces of chunk g+1 have landed: of the 12 pieces outstanding (chunks g+1, g+2, g+3) all but the 8
      // youngest are done (vector-memory operations retire in issue order; a survivor store in between only makes
      // the wait stricter)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::kWaitPieces) : "memory");
      // one barrier per chunk: every wave's pieces of chunk g+1 are in LDS, every wave has finished reading chunk g
      // (its slot is refilled below), and — first chunk of a tile — the partial sums of the previous tile are in LDS
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_chunk(nxt, static_cast<int>((g + 1) & RM));
      // The filter of the previous tile — and with it the survivor stores — runs HERE, right behind the barrier, one
      // chunk after its partial sums were collected (kDeferFilter).  A store counts in vmcnt like the LDS-DMA pieces, in
      // issue order, so one still in flight at the counted wait at the top of a chunk makes that wait ask for as many MORE
      // pieces (of chunk g+2, which nobody needs yet).  Issued at the END of an iteration — where the filter used to run —
      // a store met that wait a few hundred cycles later; issued here it has a full chunk period.  Measured A/B on one box
      // (profiles/r03/README.md): 32-query pass 0.458-0.464 ms here against 0.466-0.467 with the filter at the end; with the
      // stores stubbed 0.450.  (Adding the stores in flight to the wait's count instead — vmcnt(kWaitPieces + stores of the
      // last two iterations), valid because the count is in issue order — made the pass SLOWER, 0.479 vs 0.464: the wait's
      // slack is not what the remaining 8-14 us are; what is left is the flush itself — an LDS-counter round trip and a
      // store in ONE wave while the other seven wait for it at the next barrier, 32 times per workgroup and pass.)
      if constexpr (kDeferFilter) {
        if (ch == 1 && it > 0) stage2_finish(it - 1);
      }
      const bool finish_prev = ch == 0 && it > 0;      // the previous tile's partials were stored before this barrier
      if (finish_prev) stage2_load();
      const __amdgpu_buffer_rsrc_t rs4 = tile_rsrc(it + (ch + G::kRing) / CH);
      __builtin_amdgcn_sched_barrier(0);
      const f32x16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool skip = PARTIAL && ch == CH - 1 && 32 * w >= last_cols;   // wave-uniform (PARTIAL: see the kernel's header)
      if (PARTIAL && ch == CH - 1 && !skip && 32 * w + 32 > last_cols) {  // wave-uniform: the row ends inside this wave's slice
#pragma unroll
        for (int m = 0; m < G::kReads; ++m) {
          if (32 * w + kUnitCols * (2 * m + h) >= last_cols) cur[m] = u32x4f{0u, 0u, 0u, 0u};
        }
      }
      if (skip) {
        if (ch == 0) {                                        // a one-chunk row: this wave contributes nothing at all
          acc = zero;
          nrm_a = nrm_b = 0.f;
        }
#pragma unroll
        for (int p = 0; p < G::kPieces; ++p) issue_piece(rs4, (ch + G::kRing) % CH, static_cast<int>((g + G::kRing) & RM), p);
      } else if constexpr (kSplit) {
        // two groups of 8 columns: the cut (36 vector instructions) and six MFMAs each; the DMA pieces go out behind
        // the third and sixth MFMA of a group
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          u32x4f ah, am, al;
          split3(cur[2 * p], cur[2 * p + 1], ah, am, al);
          if constexpr (L2) {
            if (ch == 0 && p == 0) nrm_a = nrm_b = 0.f;
            add_squares(cur[2 * p]);
            add_squares(cur[2 * p + 1]);
          }
          auto mm = [&](const u32x4f& a, const u32x4f& b, bool first) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8f, a), __builtin_bit_cast(bf16x8f, b),
                                                          first ? zero : acc, 0, 0, 0);
          };
          mm(al, qf[ch][3 * p + 0], ch == 0 && p == 0);
          mm(ah, qf[ch][3 * p + 2], false);
          mm(am, qf[ch][3 * p + 1], false);
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(rs4, (ch + G::kRing) % CH, static_cast<int>((g + G::kRing) & RM), 2 * p);
          __builtin_amdgcn_sched_barrier(0);
          mm(am, qf[ch][3 * p + 0], false);
          mm(ah, qf[ch][3 * p + 1], false);
          mm(ah, qf[ch][3 * p + 0], false);
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(rs4, (ch + G::kRing) % CH, static_cast<int>((g + G::kRing) & RM), 2 * p + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int m = 0; m < G::kReads; ++m) {
          if constexpr (L2) {
            if (ch == 0 && m == 0) nrm_a = nrm_b = 0.f;
            add_squares(cur[m]);
          }
          if constexpr (BF16) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8f, cur[m]), __builtin_bit_cast(bf16x8f, qf[ch][m]),
                                                          (ch == 0 && m == 0) ? zero : acc, 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t au = cur[m][i], qu = qf[ch][m][i];    // copy the lane to a scalar before the bit cast
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, au), __builtin_bit_cast(float, qu),
                                                         (ch == 0 && m == 0 && i == 0) ? zero : acc, 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(rs4, (ch + G::kRing) % CH, static_cast<int>((g + G::kRing) & RM), m);   // kPieces == kReads
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // chunk g+1's fragments (and the previous tile's partials) are in registers
      reads_done(nxt);
      if (finish_prev) {
        stage2_pin();
        if constexpr (!kDeferFilter) stage2_finish(it - 1);
      }
#pragma unroll
      for (int m = 0; m < G::kReads; ++m) cur[m] = nxt[m];
      if (ch == CH - 1) {
        // the tile's block is complete in this wave's depth slice: hand the partials over.  With one chunk per tile
        // the other waves may still be reading the previous tile's partials (no barrier since): wait for them.
        if constexpr (CH == 1) {
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        stage1_store(acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // stores complete before the next barrier publishes them
      }
    }
  }
  if (n_my > 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stage2_load();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stage2_pin();
    stage2_finish(n_my - 1);
    if constexpr (!SAMPLE) flush_pending();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (SAMPLE) {
    float* dst = reinterpret_cast<float*>(out) + static_cast<int64_t>(r) * out_stride +
                 (static_cast<int64_t>(blockIdx.x) * 16 + 2 * w + h) * 2;
    dst[0] = mx0;
    dst[1] = mx1;
  } else {
    __syncthreads();
    if (threadIdx.x < kF32Queries) cnt[static_cast<int64_t>(blockIdx.x) * kF32Queries + threadIdx.x] = lcnt[threadIdx.x];
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
float depth_l2_margin(int dim) { return 2.f * static_cast<float>(dim) * 1.1920929e-7f; }   // 2 dim 2^-23

bool mfma_f32_path_supported(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int space) {
  // Smallest batch that takes the pass.  bf16 corpus: 2.  fp32 corpus, l2 (exact-refine mode: the select re-scores): 5.  fp32 corpus,
  // cosine (round 4): 2 — the row kernels serve 2 or 3 queries as 2 or 3 corpus passes (0.86 / 1.29 ms at 1 M x 768 against 0.50 for
  // the pass), and their four-queries-per-pass form is vector-bound on short rows (3 GB at dim 256: 0.92 ms against 0.59); only four
  // queries over rows of 768 columns or more stay on it (0.47 ms against 0.50).
  int min_q = elem_type ? kMfmaMinQueries : kMfmaF32MinQueries;
  if (!elem_type && space == DEWI_SPACE_COSINE) min_q = (n_queries == 4 && dim >= 768) ? kMfmaF32MinQueries : kMfmaMinQueries;
  if ((space != DEWI_SPACE_COSINE && space != DEWI_SPACE_L2) || n_queries < min_q || n_rows < 64 * 1024 || n_candidates > 256 || dim <= 0)
    return false;
  if (dim % kF32ChunkCols != 0) {
    // a partial last chunk (round 4): rows of whole 16-byte units (fp32 dim % 4 == 0, bf16 dim % 8 == 0) from 32 up to 1536 columns,
    // cosine; l2 over an fp32 corpus (exact-refine mode) from 132 columns = 33 units (below, the one-query kernel is
    // scan_short_rows_any, whose arithmetic the re-scoring does not repeat) to 768
    if (dim % (elem_type ? 8 : 4) != 0 || dim < 32 || dim >= 6 * kF32ChunkCols) return false;
    if (space == DEWI_SPACE_COSINE) return true;
    return elem_type == 0 && dim >= 132 && dim < kF32MaxL2Chunks * kF32ChunkCols;
  }
  const int ch = dim / kF32ChunkCols;
  if (space == DEWI_SPACE_L2) {
    if (elem_type == 0 && ch > kF32MaxL2Chunks) return false;   // fp32 rows beyond 768 columns: query pieces + row norms do not fit the registers
    return ch >= 1 && ch <= 6 && ch != 5;
  }
  // cosine, whole chunks: 256 ... 1536, 2048 (fp32: the query share of a wave is 16 registers per chunk — eight chunks fit);
  // a bf16 corpus (8 registers per chunk) also 3072 and 4096
  return (ch >= 1 && ch <= 6) || ch == 8 || (elem_type == 1 && (ch == 12 || ch == 16));
}

MfmaF32Layout plan_mfma_f32(int elem_type, int64_t n_rows, int dim, int n_queries, int n_candidates, int compute_units,
                            bool preselect) {
  MfmaF32Layout m{};
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  m.groups = (n_queries + kF32Queries - 1) / kF32Queries;
  m.q_pad = m.groups * kF32Queries;
  m.n_tiles = (n_rows + kF32TileRows - 1) / kF32TileRows;
  // Sample every `stride`-th tile: expected survivors per query ~ stride * c; keep that near 4 K (the select kernel
  // stages up to 8 K records in LDS) and the sample at no less than one tile per workgroup of the chip.
  // (preselect: thresholds two error bounds lower — a finer sample keeps the survivors of a query inside the staging)
  int64_t stride = (preselect ? 2048 : 4096) / (n_candidates > 0 ? n_candidates : 1);
  if (stride > 128) stride = 128;
  if (stride < 8) stride = 8;
  while (stride > 8 && (m.n_tiles + stride - 1) / stride < compute_units) stride /= 2;
  m.tile_stride = stride;
  m.n_sample_tiles = (m.n_tiles + stride - 1) / stride;
  m.sample_blocks = m.n_sample_tiles < compute_units ? static_cast<int>(m.n_sample_tiles) : compute_units;
  m.sample_stride = static_cast<int64_t>(m.sample_blocks) * 32;          // group maxima per query
  m.n_blocks = m.n_tiles < compute_units ? static_cast<int>(m.n_tiles) : compute_units;
  m.n_seg = m.n_blocks;
  // expected survivors per query and segment: stride * c / n_blocks; 8x head-room, at least 16 records
  int64_t cap = (8 * stride * n_candidates + m.n_seg - 1) / m.n_seg;
  m.seg_cap = static_cast<int>(cap < 16 ? 16 : cap);
  size_t off = 0;
  m.qn_off = off;      off += up(static_cast<size_t>(m.q_pad) * dim * (elem_type ? 2 : 4));
  m.qn2_off = off;     off += up(static_cast<size_t>(m.q_pad) * 4);
  m.thr_off = off;     off += up(static_cast<size_t>(m.q_pad) * 4);
  m.cnt_off = off;     off += up(static_cast<size_t>(m.groups) * m.n_seg * kF32Queries * 4);
  m.dense_off = off;   off += up(static_cast<size_t>(kF32Queries) * m.sample_stride * 4);
  m.cand_off = off;    off += up(static_cast<size_t>(m.groups) * m.n_seg * kF32Queries * m.seg_cap * 8);
  m.total = off;
  return m;
}

// PARTIAL: dim = 256 (CH - 1) + last_cols (see the kernel); otherwise dim = 256 CH and last_cols is ignored
template <bool BF16, int CH, bool L2, bool PARTIAL = false>
static hipError_t run_mfma_f32_dim(const MfmaF32Layout& m, const void* E, int64_t n_rows, int n_queries, int n_candidates,
                                   char* ws, hipStream_t stream, float thr_bias, int last_cols = kF32ChunkCols) {
  // l2 over an fp32 corpus runs in exact-refine mode (see stage2_finish and select_rerank.hip): error bound per unit of
  // ||e||^2 + ||q||^2.  bf16 corpora (opt-in, approximate) and cosine: no margin.
  const int DIM = PARTIAL ? (CH - 1) * kF32ChunkCols + last_cols : CH * kF32ChunkCols;
  const float l2_margin = L2 ? (BF16 ? 0.f : depth_l2_margin(DIM)) : thr_bias;   // the kernel's `aux`
  constexpr int kLds = depth_lds_bytes<BF16>();
  static PerDeviceOnce attr_once;   // one per instantiation
  const hipError_t ea = attr_once.run([] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_scan_f32<BF16, CH, true, L2, PARTIAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_scan_f32<BF16, CH, false, L2, PARTIAL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  });
  if (ea != hipSuccess) return ea;
  const char* qn = ws + m.qn_off;
  const float* qn2 = reinterpret_cast<const float*>(ws + m.qn2_off);
  float* thr = reinterpret_cast<float*>(ws + m.thr_off);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + m.cnt_off);
  float* dense = reinterpret_cast<float*>(ws + m.dense_off);
  uint64_t* cand = reinterpret_cast<uint64_t*>(ws + m.cand_off);
  for (int g = 0; g < m.groups; ++g) {
    const void* qg = qn + static_cast<int64_t>(g) * kF32Queries * DIM * (BF16 ? 2 : 4);
    float* tg = thr + g * kF32Queries;
    uint32_t* cg = cnt + static_cast<int64_t>(g) * m.n_seg * kF32Queries;
    uint64_t* og = cand + static_cast<int64_t>(g) * m.n_seg * kF32Queries * m.seg_cap;
    const int n_active = n_queries - g * kF32Queries < kF32Queries ? n_queries - g * kF32Queries : kF32Queries;
    // 1. group maxima over the strided sample
    const float* q2g = qn2 + g * kF32Queries;
    hipLaunchKernelGGL((mfma_scan_f32<BF16, CH, true, L2, PARTIAL>), dim3(m.sample_blocks), dim3(kF32Threads), kLds, stream, E, n_rows, qg,
                       m.n_sample_tiles, m.tile_stride, static_cast<const float*>(nullptr), reinterpret_cast<uint64_t*>(dense),
                       m.sample_stride, static_cast<uint32_t*>(nullptr), n_active, q2g, l2_margin, last_cols);
    // 2. per-query threshold: the c-th largest group maximum (real queries only)
    const hipError_t et = launch_sample_threshold(dense, m.sample_stride, m.sample_stride, n_candidates, tg, n_active, stream);
    if (et != hipSuccess) return et;
    // 3. the full pass with the filter (the kernel dewi_timing_read reports: algorithmic bytes = n_rows * dim * elem)
    timing_begin(stream);
    hipLaunchKernelGGL((mfma_scan_f32<BF16, CH, false, L2, PARTIAL>), dim3(m.n_blocks), dim3(kF32Threads), kLds, stream, E, n_rows, qg,
                       m.n_tiles, static_cast<int64_t>(1), static_cast<const float*>(tg), og, static_cast<int64_t>(m.seg_cap), cg,
                       n_active, q2g, l2_margin, last_cols);
    timing_end(stream);
  }
  return hipGetLastError();
}

hipError_t launch_mfma_f32(const MfmaF32Layout& m, int elem_type, const void* d_E, int64_t n_rows, int dim, const float* d_Q,
                           int n_queries, int n_candidates, int space, char* ws, hipStream_t stream, float thr_bias) {
  // normalised queries (fp32, or rounded to bf16 for a bf16 corpus), zero rows behind the real ones (a padding query
  // scores 0 everywhere; its threshold is forced to +inf in the kernel)
  // (l2: raw queries, rounded to bf16 for a bf16 corpus, and their squared norms)
  float* qn2 = reinterpret_cast<float*>(ws + m.qn2_off);
  hipError_t e = elem_type ? launch_prepare_queries_bf16(d_Q, reinterpret_cast<uint16_t*>(ws + m.qn_off), n_queries, m.q_pad, dim,
                                                         space, qn2, stream)
                           : launch_prepare_queries_padded(d_Q, reinterpret_cast<float*>(ws + m.qn_off), n_queries, m.q_pad, dim,
                                                           space, qn2, stream);
  if (e != hipSuccess) return e;
  const bool l2 = space == DEWI_SPACE_L2;
#define DEWI_DEPTH(CH)                                                                                                       \
  case CH:                                                                                                                   \
    if (l2) {                                                                                                                \
      if (elem_type) return run_mfma_f32_dim<true, CH, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, 0.f);          \
      if constexpr (CH <= kF32MaxL2Chunks)                                                                                   \
        return run_mfma_f32_dim<false, CH, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, 0.f);                       \
      return hipErrorInvalidValue;                                                                                           \
    }                                                                                                                        \
    return elem_type ? run_mfma_f32_dim<true, CH, false>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias)      \
                     : run_mfma_f32_dim<false, CH, false>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias);
  if (dim % kF32ChunkCols != 0) {      // a partial last chunk (mfma_f32_path_supported)
    if (dim % (elem_type ? 8 : 4) != 0) return hipErrorInvalidValue;
    const int last_cols = dim % kF32ChunkCols;
    if (l2) {                          // fp32 corpus, exact-refine mode, up to three chunks
      if (elem_type) return hipErrorInvalidValue;
      switch (dim / kF32ChunkCols + 1) {
        case 1: return run_mfma_f32_dim<false, 1, true, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, 0.f, last_cols);
        case 2: return run_mfma_f32_dim<false, 2, true, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, 0.f, last_cols);
        case 3: return run_mfma_f32_dim<false, 3, true, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, 0.f, last_cols);
        default: return hipErrorInvalidValue;
      }
    }
#define DEWI_DEPTH_PARTIAL(CH)                                                                                                  \
  case CH:                                                                                                                      \
    return elem_type ? run_mfma_f32_dim<true, CH, false, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias, last_cols) \
                     : run_mfma_f32_dim<false, CH, false, true>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias, last_cols);
    switch (dim / kF32ChunkCols + 1) {
      DEWI_DEPTH_PARTIAL(1)
      DEWI_DEPTH_PARTIAL(2)
      DEWI_DEPTH_PARTIAL(3)
      DEWI_DEPTH_PARTIAL(4)
      DEWI_DEPTH_PARTIAL(5)
      DEWI_DEPTH_PARTIAL(6)
      default: return hipErrorInvalidValue;
    }
#undef DEWI_DEPTH_PARTIAL
  }
  // cosine-only widths (round 4): 1280, 2048; bf16 corpora also 3072 and 4096
#define DEWI_DEPTH_COS(CH, ALLOW_F32)                                                                                      \
  case CH:                                                                                                                   \
    if (l2) return hipErrorInvalidValue;                                                                                     \
    if (elem_type) return run_mfma_f32_dim<true, CH, false>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias);   \
    if constexpr (ALLOW_F32) return run_mfma_f32_dim<false, CH, false>(m, d_E, n_rows, n_queries, n_candidates, ws, stream, thr_bias); \
    return hipErrorInvalidValue;
  switch (dim / kF32ChunkCols) {
    DEWI_DEPTH(1)
    DEWI_DEPTH(2)
    DEWI_DEPTH(3)
    DEWI_DEPTH(4)
    DEWI_DEPTH_COS(5, true)
    DEWI_DEPTH(6)
    DEWI_DEPTH_COS(8, true)
    DEWI_DEPTH_COS(12, false)
    DEWI_DEPTH_COS(16, false)
    default: return hipErrorInvalidValue;
  }
#undef DEWI_DEPTH_COS
#undef DEWI_DEPTH
}

}  // namespace dewi
