// Batched kNN over a bf16 corpus on the matrix cores (config C3: 1M x 768 bf16, 256 queries,
// k = 100), gfx950 (MI355X).
//
// Same contract as the scan kernels — steps 1-3 of ExactIndex.search (reference
// src/dewi/backends.py:420-444) — for up to 256 queries per corpus pass: the 256 x N score matrix
// S = Qb * Eb^T is computed tile by tile with v_mfma_f32_16x16x32_bf16 and never written; an
// epilogue keeps only scores that can still reach the top c.
//
// Roofline: co-limited on paper.  Algorithmic bytes per pass = n_rows*dim*2 (HBM, read once);
// flops = 2*256*n_rows*dim (393 GFLOP at C3) -> 256 flop/B against a ~310 flop/B machine balance.
// Measured at C3 (profiles/r02, r03): 292-333 us per pass = 4.6-5.2 TB/s and 1.2-1.35 PFLOP/s; the chip
// clocks ~1.65 GHz under this load (power), at which the matrix pipe is busy ~70 % of the time.  What
// bounds it is neither HBM nor LDS bandwidth (conflict-free ds_read_b128 at half the LDS peak) but the
// two waves of a SIMD sharing one matrix pipe and one vector-issue port under a power-limited clock:
// per 32-document tile a SIMD owes 192 MFMAs of 16 cycles = 3072 pipe cycles, and whatever the other
// wave does meanwhile (6 DMA pieces at ~60 issue cycles, the filter's compares/selects, 1-2 survivor
// stores at ~90) comes out of the same SIMD (MI355X_MICROARCH.md, "Two waves per SIMD").
// Tile delivery is the other ceiling: with 8 queries (one matrix wave per workgroup, seven waves only
// moving tiles) a pass still takes 264 us = 5.8 TB/s, the LDS-DMA fill rate of this structure.
//
// Structure (one 8-wave workgroup per CU, two waves per SIMD, persistent over 32-document tiles):
//  * QUERIES LIVE IN REGISTERS.  Wave w owns queries 32w..32w+31 for the whole kernel: their B
//    fragments (dim/32 x 2 x 4 VGPRs = 192 of the 256 available at dim 768) are loaded once, from a
//    FRAGMENT-ORDER image the preparation kernel writes (every load instruction of a wave reads one
//    contiguous KiB; a row-major image made each one touch sixteen half cache lines).
//  * DOCUMENT TILES GO THROUGH LDS BY DMA.  A tile (32 rows x dim bf16 = 48 KiB) is copied
//    global->LDS with buffer_load_dwordx4 ... lds (1 KiB per wave-instruction, no VGPR staging, non-
//    temporal) into a ring of three slots: tiles i+1 and i+2 are in flight while tile i is multiplied.
//    All 8 waves read the same tile (A operand) with ds_read_b128, two fragments ahead of the MFMA.
//  * BANK CONFLICTS: rows are 1536 B apart (= 0 mod 256 B), so an A-fragment read (16 lanes = 16
//    rows, same 16-byte column unit) would be 16-way conflicted.  The LDS image is linear (DMA
//    writes base + lane*16) and the SOURCE address is permuted instead: unit c of row r is stored
//    at unit (c & ~15) | ((c & 15) ^ (r & 15)); reads apply the same XOR -> conflict-free.
//  * A PERIOD HAS TWO HALVES PER WAVE: the matrix block (counted wait, MFMA pair, read ahead;
//    s_setprio 3, nothing else — a DMA instruction between the MFMAs stalled the in-order wave for
//    60-180 cycles each) and the side phase (wait for own pieces of the next tile: vmcnt(0),
//    everything outstanding is a period old; send the pieces of tile i+2; filter a finished tile).
//    Waves 0-3 run block(i) then side(i); waves 4-7 run side(i-1) then block(i): the two waves of a
//    SIMD are always in opposite halves.  One s_barrier per tile.
//  * FILTER.  A score passes if it is not below the query's threshold (a lower bound of its final
//    c-th best score, from a strided 1/32 sample of the corpus scanned first).  Survivors (~32c per
//    query over the whole pass) go to a quarter-segment PRIVATE to one lane: the write position is a
//    register and the record a fire-and-forget global store — no atomics.  Stores and taken branches
//    are what costs here, not compares, so a lane compacts first (branch-free select chain: first
//    passing score + pass mask) and the wave issues ONE predicated store per query half; lanes with a
//    second survivor in the same tile go round again.  History of the filter per pass: returning
//    global atomics 1.85 ms; LDS counters + per-element predication 0.54 ms; execz branch + store per
//    register 1.8 K cycles per tile and wave; 16 predicated stores without branches 1.45 K;
//    compaction ~1.1 K.  If a segment overflows (only for adversarial corpora, e.g. tens of thousands
//    of exact duplicates of a top document) its count keeps growing and the finish kernel flags the query.
//  * The 32x32x16 form of this kernel (round 1; one accumulator block per wave and half step) ran the
//    same flops 4 % slower — the chip holds a higher clock on the 16x16x32 shape — and was removed in
//    round 3; its ablations (no epilogue 305 us, no DMA after the first tile 292 us, 4 waves x 512
//    registers 357-435 us, per-wave progress flags instead of the barrier 383 vs 355 us) are in DESIGN §4.1c.
#include "select_common.hpp"

namespace dewi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4m __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kMfmaThreads = 512;     // 8 waves, two per SIMD
constexpr int kTileRows = 32;
constexpr int kQueriesPerPass = 256;  // 8 waves x 32
#ifndef DEWI_MFMA_SAMPLE_STRIDE
#define DEWI_MFMA_SAMPLE_STRIDE 32
#endif
constexpr int kSampleStride = DEWI_MFMA_SAMPLE_STRIDE;     // every 32nd tile is a sample tile
constexpr int kMaxStagedSample = 32 * 1024;  // sample scores per query the threshold kernel keeps in LDS (128 KiB)
constexpr int kTileBufs = 3;          // LDS ring: the tile being multiplied + two tiles of DMA in flight
#ifndef DEWI_MFMA_DMA_AUX
#define DEWI_MFMA_DMA_AUX 2   // cache policy bits of the tile DMA (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  The corpus is read once:
                               // non-temporal 334 us per pass at 256 queries and 240 us at 8, default policy 355 and 264
#endif
#ifndef DEWI_MFMA_PRIO
#define DEWI_MFMA_PRIO 1
#endif
constexpr int kSegPerBlock = 4;       // lane-private survivor quarter-segments per workgroup and query

using GlobalPtr = const void __attribute__((address_space(1)))*;
using LdsPtr = void __attribute__((address_space(3)))*;

// Query preparation: normalise (cosine, unless the norm is 0) and round to bf16; rows >= n_queries
// of the 32-padded block are zero.  One wave per query row.
// qn2 (optional): ||bf16-rounded prepared query||^2 per output row (float64-summed, one rounding), for the l2 score of
// the depth-split pass.
// frag_order != 0 (dim % 32 == 0; the 256-query kernel's input): the image is written in the order mfma_scan_bf16_s16
// loads its B fragments instead of row-major — the 8 columns 8j..8j+7 of query q (k-step s = j / 4, lane group g = j % 4)
// go to 16-byte unit (((q / 32) * 2 + (q / 16) % 2) * (dim / 32) + s) * 64 + 16 g + q % 16 — so that each of a wave's
// fragment loads reads one contiguous KiB.
__device__ __forceinline__ int64_t fragment_unit(int q, int j, int dim) {
  return (static_cast<int64_t>((q >> 5) * 2 + ((q >> 4) & 1)) * (dim >> 5) + (j >> 2)) * 64 + 16 * (j & 3) + (q & 15);
}

__global__ __launch_bounds__(kWave) void prepare_queries_bf16(const float* __restrict__ Q, uint16_t* __restrict__ Qb,
                                                              int n_queries, int dim, int space, float* __restrict__ qn2,
                                                              int frag_order) {
  const int lane = lane_id();
  const int row = static_cast<int>(blockIdx.x);
  uint16_t* o = Qb + static_cast<int64_t>(row) * dim;
  if (row >= n_queries) {
    if (frag_order) {
      for (int j = lane; j < (dim >> 3); j += kWave)
        reinterpret_cast<uint4*>(Qb)[fragment_unit(row, j, dim)] = uint4{0u, 0u, 0u, 0u};
    } else {
      for (int j = lane; j < dim; j += kWave) o[j] = 0;
    }
    if (qn2 != nullptr && lane == 0) qn2[row] = 0.f;
    return;
  }
  auto widen = [](uint16_t b) { return __uint_as_float(static_cast<uint32_t>(b) << 16); };
  const float* q = Q + static_cast<int64_t>(row) * dim;
  auto to_bf16 = [](float v) {
    const uint32_t u = __float_as_uint(v);
    return (v != v) ? static_cast<uint16_t>((u >> 16) | 0x0040u) : static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  };
  // dim % 4 == 0 and at most 2048 (every matrix-core path): the row stays in registers — all of its 16-byte loads
  // are in flight together, one pass for the norm, one for the division, 8-byte stores (the one-element-at-a-time
  // loops below cost 7-8 us per launch in load round trips for a 768 KB job)
  if ((dim & 3) == 0 && dim <= 2048) {
    typedef float f32x4p __attribute__((ext_vector_type(4)));
    typedef uint16_t u16x4p __attribute__((ext_vector_type(4)));
    const int n4 = dim >> 2;
    f32x4p v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + u * kWave;
      v[u] = j < n4 ? reinterpret_cast<const f32x4p*>(q)[j] : f32x4p{0.f, 0.f, 0.f, 0.f};
    }
    float norm = 1.f;
    bool scale = false;
    if (space == DEWI_SPACE_COSINE) {
      double ss = 0.0;   // float64: the same norm as every other kernel's (common.hpp, wave_query_norm)
#pragma unroll
      for (int u = 0; u < 8; ++u) ss += square_f64(v[u].x) + square_f64(v[u].y) + square_f64(v[u].z) + square_f64(v[u].w);
      norm = wave_query_norm(ss);
      scale = norm > 0.f;
    }
    double out2 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lane + u * kWave;
      if (j < n4) {
        u16x4p r;
        r.x = to_bf16(scale ? __fdiv_rn(v[u].x, norm) : v[u].x);
        r.y = to_bf16(scale ? __fdiv_rn(v[u].y, norm) : v[u].y);
        r.z = to_bf16(scale ? __fdiv_rn(v[u].z, norm) : v[u].z);
        r.w = to_bf16(scale ? __fdiv_rn(v[u].w, norm) : v[u].w);
        out2 += square_f64(widen(r.x)) + square_f64(widen(r.y)) + square_f64(widen(r.z)) + square_f64(widen(r.w));
        if (frag_order)   // columns 4j..4j+3 = half (j & 1) of the 8-column group j >> 1
          reinterpret_cast<u16x4p*>(Qb)[fragment_unit(row, j >> 1, dim) * 2 + (j & 1)] = r;
        else
          reinterpret_cast<u16x4p*>(o)[j] = r;
      }
    }
    if (qn2 != nullptr) {
      out2 = wave_sum_f64(out2);
      if (lane == 0) qn2[row] = static_cast<float>(out2);
    }
    return;
  }
  // (any other dim: row-major only — the fragment order is asked for by the 256-query kernel alone, dim % 128 == 0)
  float norm = 1.f;
  bool scale = false;
  if (space == DEWI_SPACE_COSINE) {
    double ss = 0.0;
    for (int j = lane; j < dim; j += kWave) ss += square_f64(q[j]);
    norm = wave_query_norm(ss);
    scale = norm > 0.f;
  }
  double out2 = 0.0;
  for (int j = lane; j < dim; j += kWave) {
    const uint16_t r = to_bf16(scale ? __fdiv_rn(q[j], norm) : q[j]);
    out2 += square_f64(widen(r));
    o[j] = r;
  }
  if (qn2 != nullptr) {
    out2 = wave_sum_f64(out2);
    if (lane == 0) qn2[row] = static_cast<float>(out2);
  }
}

// ---------------------------------------------------------------------------------------------
// The pass itself, on v_mfma_f32_16x16x32_bf16: per 32 columns four 16-cycle MFMAs (2 document halves x 2 query
// halves).  Fragment geometry:
//   lane l = (c = l & 15, g = l >> 4).  k-step s covers columns 32 s .. 32 s + 31; lane group g holds columns 32 s + 8 g .. +7.
//   B (queries, registers): qf[qh][s] = Qb[32 wave + 16 qh + c][32 s + 8 g .. +7], qh = 0, 1        (2 x 24 x 4 = 192 VGPRs at dim 768)
//   A (documents, LDS):     fragment f = 2 s + dh = rows 16 dh + c of the tile, 16-byte unit 4 s + g, swizzled as the DMA
//                           wrote it: unit (u & ~15) | ((u & 15) ^ (row & 15)) — conflict-free for the 4 x 16-lane
//                           groups of ds_read_b128 (rows {0-3, 12-15} of group g with rows {4-11} of group g+1).
//   D:                      acc[dh][qh][t] = score of document 16 dh + 4 g + t against query 32 wave + 16 qh + c.
// A lane therefore holds TWO queries (qh = 0, 1) with eight documents each; a query belongs to four lanes (g = 0..3) of
// one wave: quarter-segments, segment index 4 blockIdx.x + g, two slot counters per lane.
// KS = dim / 16 (dim % 128 == 0, dim <= 768).  Tiles handled by a launch: t = first_tile + i * tile_stride.
// ---------------------------------------------------------------------------------------------
typedef float f32x4a __attribute__((ext_vector_type(4)));

template <int KS, bool DENSE>
__global__ __launch_bounds__(512, 2) void mfma_scan_bf16_s16(
    const uint16_t* __restrict__ E, int64_t n_rows, const uint16_t* __restrict__ Qb, int64_t n_tiles,
    int64_t tile_stride, const float* __restrict__ thr, uint64_t* __restrict__ out, int64_t out_stride,
    uint32_t* __restrict__ cnt, int n_active, float thr_bias) {
  // DENSE (sample pass): out is a float array: out[q * out_stride + (4 * blockIdx.x + g) * 8 + 4 dh + t] = the best score
  //        that accumulator register saw over this workgroup's (strided) tiles — 32 group maxima per workgroup and query.
  // filter: raw records out[((4 * blockIdx.x + g) * 256 + q) * out_stride + slot], cnt[q * 4 gridDim.x + 4 * blockIdx.x + g].
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int DIM = KS * 16;
  constexpr int KS2 = KS / 2;                      // 32-column k-steps
  constexpr int NF = KS;                           // A fragments per tile (2 per k-step)
  constexpr int UPR = DIM / 8;                     // 16-byte units per row
  constexpr int TILE_BYTES = kTileRows * DIM * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int NW = 8;
  constexpr int PPW = PIECES / NW;
  static_assert(KS % 8 == 0 && KS <= 48, "dim must be a multiple of 128, at most 768");
  extern __shared__ __attribute__((aligned(16))) char lds[];  // kTileBufs x TILE_BYTES

  const int lane = lane_id();
  const int wave = static_cast<int>(threadIdx.x) >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool active = 32 * wave_u < n_active;

  // Qb is the FRAGMENT-ORDER image prepare_queries_bf16 writes (frag_order): unit ((2 wave + qh) * KS2 + s) * 64 + lane
  // holds Qb[32 wave + 16 qh + c][32 s + 8 g .. +7] — one contiguous KiB per load instruction of the wave
  bf16x8 qf[2][KS2];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const bf16x8* qp = reinterpret_cast<const bf16x8*>(Qb) + static_cast<int64_t>(2 * wave + qh) * KS2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS2; ++s) qf[qh][s] = qp[active ? 64 * s : 0];
  }
  float thr_l[2];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh) {
    const int q = 32 * wave + 16 * qh + c;
    // thr_bias: 0, or twice the error bound of these scores when they only PRE-SELECT for an exact re-scoring (an fp32
    // corpus scanned through its bf16 shadow: launch.hpp, shadow_margin)
    thr_l[qh] = DENSE ? -__builtin_inff() : (q < n_active ? thr[q] - thr_bias : __builtin_inff());
  }
#pragma unroll
  for (int s = 0; s < KS2; ++s) {
    asm volatile("" ::"v"(qf[0][s]));
    asm volatile("" ::"v"(qf[1][s]));
  }
  asm volatile("" ::"v"(thr_l[0]), "v"(thr_l[1]));

  // ---- per-lane DMA source offsets (bytes from the tile's first row).  The swizzle repeats every 16 rows (= KS/2
  // pieces), so piece i and piece i + VO differ by the constant 16 * DIM * 2 bytes, which goes into the instruction's
  // scalar offset: VO = KS/(2 NW) registers instead of PPW = KS/NW.
  constexpr int VO = (KS % (2 * NW) == 0) ? KS / (2 * NW) : PPW;
  uint32_t voff[VO];
#pragma unroll
  for (int i = 0; i < VO; ++i) {
    const int piece = i * NW + wave;
    const int x = piece * 64 + lane;
    const int row = x / UPR, cp = x % UPR;
    const int cc = (cp & ~15) | ((cp & 15) ^ (row & 15));
    voff[i] = static_cast<uint32_t>(row * (DIM * 2) + cc * 16);
  }
  // ---- A-fragment read addresses: row c (+ 16 dh as an immediate), unit 4 s + g -> low four bits 4 (s & 3) + g, XOR c
  uint32_t a_addr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    a_addr[j] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsPtr)(lds))) +
                static_cast<uint32_t>(c * UPR * 16 + 16 * ((4 * j + g) ^ c));

  const char* Eb = reinterpret_cast<const char*>(E);
  auto tile_rsrc = [&](int64_t tile_index) {
    const int64_t row0 = tile_index * kTileRows;
    const int64_t left = n_rows - row0;
    const int valid_rows = left < kTileRows ? static_cast<int>(left) : kTileRows;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Eb + row0 * (DIM * 2)), 0, valid_rows * (DIM * 2),
                                             0x00020000);
  };
  auto issue_piece = [&](__amdgpu_buffer_rsrc_t rsrc, int buf, int i) {
    char* l = lds + buf * TILE_BYTES + (i * NW + wave_u) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)(l), 16, voff[i % VO], (i / VO) * (16 * DIM * 2), 0, DEWI_MFMA_DMA_AUX);
  };

  // ---- survivor quarter-segments: byte offset of the next free record, one per query half
  uint32_t off[2];
#pragma unroll
  for (int qh = 0; qh < 2; ++qh)
    off[qh] = ((static_cast<uint32_t>(blockIdx.x) * 4 + g) * kQueriesPerPass + (32 * wave + 16 * qh + c)) *
              static_cast<uint32_t>(out_stride) * 8u;

  const int64_t first = static_cast<int64_t>(blockIdx.x);
  const int64_t step = static_cast<int64_t>(gridDim.x);
  if (first < n_tiles) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(first * tile_stride);
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(rs, 0, i);
  }
  if (first + step < n_tiles) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc((first + step) * tile_stride);
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(rs, 1, i);
  }
  const uint64_t out_bits = reinterpret_cast<uint64_t>(out);
  uint64_t* const out_uniform = reinterpret_cast<uint64_t*>(
      (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(out_bits >> 32)))) << 32) |
      static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(out_bits))));

  float mx[DENSE ? 2 : 1][8];
  if constexpr (DENSE) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
#pragma unroll
      for (int j = 0; j < 8; ++j) mx[qh][j] = -__builtin_inff();
    }
  }
  // The epilogue of tile i.  acc[dh][qh][t]: document row0 + 16 dh + 4 g + t, query 32 wave + 16 qh + c.
  auto epilogue = [&](f32x4a (&acc)[2][2], int64_t i) {
    const int64_t row0 = i * tile_stride * kTileRows;
    const int doc0 = static_cast<int>(row0) + 4 * g;
    if (row0 + kTileRows > n_rows) {                           // partial last tile: padding rows never pass
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (row0 + 16 * dh + 4 * g + t >= n_rows) {
            acc[dh][0][t] = -__builtin_inff();
            acc[dh][1][t] = -__builtin_inff();
          }
        }
      }
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      // the eight scores of this lane for query half qh, in document order: j = 4 dh + t -> row offset 16 dh + t.
      // One compaction chain and one predicated store per query half: a single 16-register chain with a selected
      // threshold / segment per register measured slower (more second rounds, more vector instructions): 0.975 vs
      // 0.957 of the 32x32x16 kernel's time.
#define sc_(j) acc[(j) >> 2][qh][(j) & 3]
      if constexpr (DENSE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) mx[qh][j] = __builtin_fmaxf(mx[qh][j], sc_(j));
      } else {
        const uint32_t seg_q = (static_cast<uint32_t>(blockIdx.x) * 4 + g) * kQueriesPerPass + (32 * wave + 16 * qh + c);
        const uint32_t end_bytes = (seg_q + 1u) * static_cast<uint32_t>(out_stride) * 8u;
        const bool tight = off[qh] + 8u * 8u > end_bytes;   // also true once the segment has overflowed
        if (__builtin_amdgcn_ballot_w64(tight) == 0ull) {
          auto store_where = [&](uint32_t flag, uint32_t row_off, float score) {
            const uint64_t rec = (static_cast<uint64_t>(static_cast<uint32_t>(doc0) + row_off) << 32) | __float_as_uint(score);
            uint64_t saved_exec;
            asm volatile(
                "v_cmp_ne_u32 vcc, 0, %[f]\n\t"
                "s_and_saveexec_b64 %[sv], vcc\n\t"
                "global_store_dwordx2 %[off], %[rec], %[base]\n\t"
                "v_add_u32 %[off], 8, %[off]\n\t"
                "s_mov_b64 exec, %[sv]"
                : [off] "+v"(off[qh]), [sv] "=&s"(saved_exec)
                : [f] "v"(flag), [rec] "v"(rec), [base] "s"(out_uniform)
                : "vcc", "memory");
          };
          uint32_t mask = 0;
          float firstv = 0.f;
#pragma unroll
          for (int j = 7; j >= 0; --j) {
            const bool pass = !(sc_(j) < thr_l[qh]);            // NaN passes (NumPy ranks NaN first)
            firstv = pass ? sc_(j) : firstv;
            mask = mask + mask + (pass ? 1u : 0u);
          }
          uint32_t n_pass = static_cast<uint32_t>(__builtin_popcount(mask));
          const uint32_t j0 = static_cast<uint32_t>(__builtin_ctz(mask | 0x100u));   // first passing register (8: none)
          uint32_t ro = (j0 & 3u) + 16u * (j0 >> 2);
          store_where(n_pass, ro, firstv);
          uint32_t more = n_pass > 1u ? 1u : 0u;
          while (__builtin_amdgcn_ballot_w64(more != 0u) != 0ull) {
            uint32_t ro_next = 0;
            float next = 0.f;
#pragma unroll
            for (int j = 7; j >= 1; --j) {
              const uint32_t roj = static_cast<uint32_t>((j & 3) + 16 * (j >> 2));
              const bool pass = !(sc_(j) < thr_l[qh]) && roj > ro;
              next = pass ? sc_(j) : next;
              ro_next = pass ? roj : ro_next;
            }
            store_where(more, ro_next, next);
            ro = more ? ro_next : ro;
            n_pass -= more;
            more = n_pass > 1u ? 1u : 0u;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool pass = !(sc_(j) < thr_l[qh]);
            if (pass) {
              if (off[qh] < end_bytes)
                out[off[qh] >> 3] = (static_cast<uint64_t>(static_cast<uint32_t>(doc0 + (j & 3) + 16 * (j >> 2))) << 32) |
                                    __float_as_uint(sc_(j));
              off[qh] += 8u;   // keeps counting past the end: the finish kernel sees count > capacity
            }
          }
        }
      }
#undef sc_
    }
  };

  const bool deferred = wave_u >= NW / 2;
  f32x4a acc[2][2];
  int64_t prev = -1;
  int buf = 0;
  if (first + step < n_tiles) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (int64_t i = first; i < n_tiles; i += step) {
    const bool has_next2 = i + 2 * step < n_tiles;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int buf2 = buf + 2 >= kTileBufs ? buf + 2 - kTileBufs : buf + 2;

    constexpr int kAhead = 2;   // fragments in flight ahead of the MFMAs (register budget: 256 with 192 of them queries)
    u32x4m ring[kAhead + 1];
    // fragment f = 2 s + dh: address register s & 3, immediate 256 (s >> 2) bytes + 16 rows for dh = 1
    auto read_fragment = [&](u32x4m& dst, int f) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a_addr[(f >> 1) & 3]), "n"(256 * (f >> 3) + (f & 1) * 16 * UPR * 16));
    };
#pragma unroll
    for (int f = 0; f < kAhead && active; ++f) read_fragment(ring[f], f);

    auto side_phase = [&](bool filter, int64_t filter_tile) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (has_next2) {
        const __amdgpu_buffer_rsrc_t next_rsrc = tile_rsrc((i + 2 * step) * tile_stride);
#pragma unroll
        for (int p = 0; p < PPW; ++p) issue_piece(next_rsrc, buf2, p);
      }
      if (filter && active) epilogue(acc, filter_tile);
    };

    if (deferred) side_phase(prev >= 0, prev);
    if (active) {
      if (DEWI_MFMA_PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        u32x4m& cur = ring[f % (kAhead + 1)];
        if (f + kAhead < NF) read_fragment(ring[(f + kAhead) % (kAhead + 1)], f + kAhead);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"((NF - 1 - f) < kAhead ? (NF - 1 - f) : kAhead));
        const bf16x8 a = __builtin_bit_cast(bf16x8, cur);
        const f32x4a zero = {0.f, 0.f, 0.f, 0.f};
        constexpr int dummy = 0;
        (void)dummy;
        const int s = f >> 1, dh = f & 1;
        acc[dh][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[0][s], s == 0 ? zero : acc[dh][0], 0, 0, 0);
        acc[dh][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[1][s], s == 0 ? zero : acc[dh][1], 0, 0, 0);
      }
      if (DEWI_MFMA_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (!deferred) side_phase(true, i);
    prev = i;
#pragma unroll
    for (int j = 0; j < 4; ++j) a_addr[j] = buf == kTileBufs - 1 ? a_addr[j] - (kTileBufs - 1) * TILE_BYTES : a_addr[j] + TILE_BYTES;
    buf = buf == kTileBufs - 1 ? 0 : buf + 1;
  }
  if (deferred && prev >= 0 && active) epilogue(acc, prev);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (!DENSE) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      const uint32_t seg = static_cast<uint32_t>(blockIdx.x) * 4 + g, q = 32 * wave + 16 * qh + c;
      const uint32_t seg_q = seg * kQueriesPerPass + q;
      // counts are query-major, cnt[q][segment]: the select kernel reads a query's 4 * gridDim counts in one sweep
      cnt[q * (4u * gridDim.x) + seg] = (off[qh] >> 3) - seg_q * static_cast<uint32_t>(out_stride);
    }
  } else if (active) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      f32x4v* dst = reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(out) +
                                              static_cast<int64_t>(32 * wave + 16 * qh + c) * out_stride +
                                              (static_cast<int64_t>(blockIdx.x) * 4 + g) * 8);
      dst[0] = f32x4v{mx[qh][0], mx[qh][1], mx[qh][2], mx[qh][3]};
      dst[1] = f32x4v{mx[qh][4], mx[qh][5], mx[qh][6], mx[qh][7]};
    }
  }
#endif
}

// Per-query threshold from the dense sample scores: the score of the c-th best sample document is a
// lower bound of the query's final c-th best score.  One workgroup per query; exact 3-pass MSB radix
// select (11 + 11 + 10 bits) on the order-preserving keys of the n_sample fp32 scores.
//
// The three passes are bound by reading the scores (256 queries x 31 K samples = 32 MB per pass,
// more than the L2s hold): STAGED keeps the query's keys in LDS (n_sample <= 32 K: 125 KiB of the
// 160) during the first pass — read from global once, with 16-byte loads — and passes 2 and 3 run
// out of LDS.  (Tried and dropped: a floor from the c-th largest per-thread maximum to thin out the
// first-pass LDS atomics — the extra read pass cost more than the atomics it saved, 53 vs 42 us.)
struct RadixPick {
  uint32_t prefix, remaining;
  bool short_input;   // fewer keys than the wanted rank
};

// One radix pass over the histogram in LDS: finds the digit holding the `remaining`-th largest key.
// Every thread returns the same values.  `hist` holds kBins counters.
template <int kBins>
__device__ __forceinline__ void pick_digit_desc(const uint32_t* hist, uint32_t* wave_tot, uint32_t* pick, int bits,
                                                RadixPick& st) {
  const int tid = static_cast<int>(threadIdx.x);
  const int lane = tid & 63, wave = tid >> 6, n_waves = static_cast<int>(blockDim.x) >> 6;
  // suffix sums: thread t owns bins 2t and 2t+1
  const uint32_t h0 = 2 * tid < kBins ? hist[2 * tid] : 0u, h1 = 2 * tid + 1 < kBins ? hist[2 * tid + 1] : 0u;
  uint32_t sfx = h0 + h1;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_down(sfx, off, kWave);
    if (lane + off < 64) sfx += o;
  }
  if (lane == 0) wave_tot[wave] = sfx;
  __syncthreads();
  uint32_t above = 0;
  for (int w = wave + 1; w < n_waves; ++w) above += wave_tot[w];
  const uint32_t incl1 = sfx + above - h0;  // keys with digit >= 2t+1
  const uint32_t incl0 = sfx + above;       // keys with digit >= 2t
  if (tid == 0) pick[2] = incl0;            // total
  if (incl1 >= st.remaining && incl1 - h1 < st.remaining) {
    pick[0] = 2 * tid + 1;
    pick[1] = incl1 - h1;
  } else if (incl0 >= st.remaining && incl1 < st.remaining) {
    pick[0] = 2 * tid;
    pick[1] = incl1;
  }
  __syncthreads();
  if (pick[2] < st.remaining) {
    st.short_input = true;
  } else {
    st.prefix = (st.prefix << bits) | pick[0];
    st.remaining -= pick[1];
  }
  __syncthreads();
}

// n_sample % 4 == 0 and 16-byte aligned rows (the dense pass writes whole 32-document tiles).
template <bool STAGED>
__global__ __launch_bounds__(kSelectThreads) void sample_threshold_kernel(const float* __restrict__ dense,
                                                                          int64_t n_sample, int64_t stride,
                                                                          int n_candidates, float* __restrict__ thr) {
  constexpr int kBins = 2048;
  extern __shared__ __attribute__((aligned(16))) uint32_t staged[];   // STAGED: n_sample keys
  __shared__ uint32_t hist[kBins];
  __shared__ uint32_t wave_tot[kSelectThreads / kWave];
  __shared__ uint32_t pick[3];
  typedef float f32x4s __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const f32x4s* s4 = reinterpret_cast<const f32x4s*>(dense + static_cast<int64_t>(blockIdx.x) * stride);
  const int n4 = static_cast<int>(n_sample >> 2);
  // Two radix passes of up to 11 bits over the keys' offsets from the SMALLEST key of the query (block minimum and
  // maximum first): real scores crowd into a narrow range — cosines of one corpus, or l2 scores around -||q||^2 — and
  // the top bits of the raw key are then the same for all 8192 values: every LDS atomic of a pass hit one bin and
  // serialised (13 us instead of 7 on l2 batches).  The threshold is the LOWER EDGE of the bin the second pass ends in:
  // exact when the span is below 2^22, else a valid lower bound of the c-th best score, looser by less than 2^-11 of
  // the span (a few more survivors in a few thousand); a third pass that made it exact cost 2.5 us per batch.
  __shared__ uint32_t wave_min[kSelectThreads / kWave], wave_max[kSelectThreads / kWave];
  // up to 8 keys per thread (8192 group maxima: every launch of this library) stay in registers for both passes;
  // longer inputs go through the staged LDS copy / are read again
  constexpr int kRegIter = 2;
  const bool in_regs = n4 <= kRegIter * nt;
  u32x4s kreg[kRegIter];
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kRegIter; ++u) {
      const int i = tid + u * nt;
      f32x4s v = {0.f, 0.f, 0.f, 0.f};
      if (i < n4) v = s4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kreg[u][e] = ord_f32(v[e]);
        if (i < n4) {
          kmin = kreg[u][e] < kmin ? kreg[u][e] : kmin;
          kmax = kreg[u][e] > kmax ? kreg[u][e] : kmax;
        }
      }
    }
  } else {
    for (int i = tid; i < n4; i += nt) {
      const f32x4s v = s4[i];
      u32x4s key;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        key[e] = ord_f32(v[e]);
        kmin = key[e] < kmin ? key[e] : kmin;
        kmax = key[e] > kmax ? key[e] : kmax;
      }
      if (STAGED) reinterpret_cast<u32x4s*>(staged)[i] = key;
    }
  }
  kmin = wave_min_u32(kmin);
  kmax = wave_max_u32(kmax);
  if ((tid & 63) == 0) {
    wave_min[tid >> 6] = kmin;
    wave_max[tid >> 6] = kmax;
  }
  for (int b = tid; b < kBins; b += nt) hist[b] = 0;
  __syncthreads();
  for (int w = 0; w < nt / kWave; ++w) {
    kmin = wave_min[w] < kmin ? wave_min[w] : kmin;
    kmax = wave_max[w] > kmax ? wave_max[w] : kmax;
  }
  const uint32_t base = kmin, span = kmax - kmin;
  const int total_bits = span ? 32 - __builtin_clz(span) : 1;
  const int shift1 = total_bits > 11 ? total_bits - 11 : 0;
  const int shift2 = shift1 > 11 ? shift1 - 11 : 0;
  RadixPick st{0u, static_cast<uint32_t>(n_candidates), false};
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && shift1 == 0) break;            // the first pass resolved every bit of the span
    const int bits = pass == 0 ? 11 : shift1 - shift2;
    const int shift = pass == 0 ? shift1 : shift2;
    if (pass > 0) {                                 // (the first pass's histogram was cleared before the barrier above)
      for (int b = tid; b < kBins; b += nt) hist[b] = 0;
      __syncthreads();
    }
    auto count = [&](const u32x4s& key) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t off = key[e] - base;
        if (pass == 0 || (off >> shift1) == st.prefix) atomicAdd(&hist[(off >> shift) & ((1u << bits) - 1u)], 1u);
      }
    };
    if (in_regs) {
#pragma unroll
      for (int u = 0; u < kRegIter; ++u)
        if (tid + u * nt < n4) count(kreg[u]);
    } else {
#pragma unroll 2
      for (int i = tid; i < n4; i += nt) {
        u32x4s key;
        if (STAGED) {
          key = reinterpret_cast<const u32x4s*>(staged)[i];
        } else {
          const f32x4s v = s4[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) key[e] = ord_f32(v[e]);
        }
        count(key);
      }
    }
    __syncthreads();
    pick_digit_desc<kBins>(hist, wave_tot, pick, bits, st);
    if (st.short_input) {  // fewer sample scores than candidates (first pass only): no bound
      if (tid == 0) thr[blockIdx.x] = -__builtin_inff();
      return;
    }
  }
  if (tid == 0) thr[blockIdx.x] = unord_f32(base + (st.prefix << (shift1 == 0 ? 0 : shift2)));
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// |<e,q> computed from bf16(e), bf16(q) on the matrix cores - <e,q> computed by the fp32 row kernels| for rows and queries
// of norm <= 1.0001: bf16 rounding (2^-9 relative per factor: (1 + 2^-9)^2 - 1 of sum |e_i q_i| <= ||e|| ||q||), dim fp32
// accumulations here (2^-23 each, conservatively) and dim there (2^-24).
float shadow_margin(int dim) { return (0.00390625f * 1.002f + 1.5f * static_cast<float>(dim) * 1.1920929e-7f) * 1.0003f; }

bool mfma_path_supported(int64_t n_rows, int dim, int n_queries, int n_candidates, int space) {
  return space == DEWI_SPACE_COSINE && n_queries >= kMfmaMinQueries && dim % 128 == 0 && dim <= 768 &&
         n_rows >= 64 * 1024 && n_candidates <= kMaxSortCandidates &&
         n_rows / (kTileRows * kSampleStride) * kTileRows >= 4 * static_cast<int64_t>(n_candidates);
}

MfmaLayout plan_mfma(int64_t n_rows, int dim, int n_queries, int n_candidates, int compute_units, bool preselect) {
  MfmaLayout m{};
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  m.groups = (n_queries + kQueriesPerPass - 1) / kQueriesPerPass;
  m.q_pad = m.groups * kQueriesPerPass;
  m.n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  // Pre-selection for an exact re-scoring lowers every threshold by two error bounds (~0.008 in cosine at dim 768): at c = 200
  // that alone adds ~5 K survivors per query behind a 1/32 sample.  A finer sample puts the threshold higher up the tail,
  // where the same band holds fewer rows: ~6 K survivors at c = 200 with 1/16 (1 M x 768 gaussian rows), ~7.5 K at c = 500 with 1/8.
  m.tile_stride = !preselect || n_candidates <= 64 ? kSampleStride : (n_candidates <= 256 ? kSampleStride / 2 : kSampleStride / 4);
  m.n_sample_tiles = (m.n_tiles + m.tile_stride - 1) / m.tile_stride;
  // group maxima per query: sample workgroups x 2 lane halves x 16 accumulator registers
  const int64_t sample_blocks = m.n_sample_tiles < compute_units ? m.n_sample_tiles : compute_units;
  m.sample_stride = sample_blocks * 32;
  // Filter pass: one workgroup per CU; every (workgroup, lane half, query) has a private half-segment.
  // Expected survivors per query ~ n_rows * c / n_sample = 32 c, spread evenly over the half-segments;
  // 4x head-room, at least 32 records.
  m.n_blocks = m.n_tiles < compute_units ? static_cast<int>(m.n_tiles) : compute_units;
  m.n_seg = kSegPerBlock * m.n_blocks;
  int64_t cap = (4ll * kSampleStride * n_candidates + m.n_seg - 1) / m.n_seg;   // (sized for the 1/32 sample: head-room for the band)
  m.seg_cap = static_cast<int>(cap < 32 ? 32 : cap);
  size_t off = 0;
  m.qb_off = off;      off += up(static_cast<size_t>(m.q_pad) * dim * 2);
  m.thr_off = off;     off += up(static_cast<size_t>(m.q_pad) * 4);
  m.cnt_off = off;     off += up(static_cast<size_t>(m.groups) * m.n_seg * kQueriesPerPass * 4);
  m.dense_off = off;   off += up(static_cast<size_t>(kQueriesPerPass) * m.sample_stride * 4);
  m.cand_off = off;    off += up(static_cast<size_t>(m.groups) * m.n_seg * kQueriesPerPass * m.seg_cap * 8);
  m.total = off;
  return m;
}

// Thresholds of `n_queries` queries from their sample values (dense [n_queries][stride], n_sample valid
// per query): shared by the bf16 and the fp32 matrix-core paths.
hipError_t launch_sample_threshold(const float* dense, int64_t n_sample, int64_t stride, int n_candidates, float* thr,
                                   int n_queries, hipStream_t stream) {
  static PerDeviceOnce once;
  const hipError_t e = once.run([] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_threshold_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kMaxStagedSample * 4);
  });
  if (e != hipSuccess) return e;
  if (n_sample <= kMaxStagedSample)
    hipLaunchKernelGGL(sample_threshold_kernel<true>, dim3(n_queries), dim3(kSelectThreads),
                       static_cast<size_t>(n_sample) * 4, stream, dense, n_sample, stride, n_candidates, thr);
  else
    hipLaunchKernelGGL(sample_threshold_kernel<false>, dim3(n_queries), dim3(kSelectThreads), 0, stream, dense, n_sample,
                       stride, n_candidates, thr);
  return hipGetLastError();
}

using ScanKernel = void (*)(const uint16_t*, int64_t, const uint16_t*, int64_t, int64_t, const float*, uint64_t*, int64_t, uint32_t*,
                            int, float);
template <int KS, bool DENSE>
static ScanKernel scan_kernel() {
  return &mfma_scan_bf16_s16<KS, DENSE>;
}

template <int KS>
static hipError_t run_mfma_dim(const MfmaLayout& m, const uint16_t* E, int64_t n_rows, int n_queries, int n_candidates,
                               char* ws, int compute_units, hipStream_t stream, float thr_bias) {
  constexpr int DIM = KS * 16;
  const int lds_bytes = kTileBufs * kTileRows * DIM * 2;
  static PerDeviceOnce attr_once;   // one per KS instantiation
  const hipError_t ea = attr_once.run([] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<KS, true>()),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(scan_kernel<KS, false>()),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  });
  if (ea != hipSuccess) return ea;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(ws + m.qb_off);
  float* thr = reinterpret_cast<float*>(ws + m.thr_off);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(ws + m.cnt_off);
  float* dense = reinterpret_cast<float*>(ws + m.dense_off);
  uint64_t* cand = reinterpret_cast<uint64_t*>(ws + m.cand_off);
  for (int g = 0; g < m.groups; ++g) {
    const uint16_t* qg = qb + static_cast<int64_t>(g) * kQueriesPerPass * DIM;
    float* tg = thr + g * kQueriesPerPass;
    uint32_t* cg = cnt + static_cast<int64_t>(g) * m.n_seg * kQueriesPerPass;
    uint64_t* og = cand + static_cast<int64_t>(g) * m.n_seg * kQueriesPerPass * m.seg_cap;
    const int n_active = n_queries - g * kQueriesPerPass < kQueriesPerPass ? n_queries - g * kQueriesPerPass : kQueriesPerPass;
    // 1. dense scores of the strided sample
    const int sample_blocks = m.n_sample_tiles < compute_units ? static_cast<int>(m.n_sample_tiles) : compute_units;
    const ScanKernel k_sample = scan_kernel<KS, true>(), k_filter = scan_kernel<KS, false>();
    hipLaunchKernelGGL(k_sample, dim3(sample_blocks), dim3(kMfmaThreads), lds_bytes, stream, E, n_rows,
                       qg, m.n_sample_tiles, static_cast<int64_t>(m.tile_stride), static_cast<const float*>(nullptr),
                       reinterpret_cast<uint64_t*>(dense), m.sample_stride, static_cast<uint32_t*>(nullptr), n_active, 0.f);
    // 2. per-query threshold (real queries only: a padding query's sample scores are all equal, which is the
    //    worst case of the histogram select, and its threshold is not used)
    const hipError_t et = launch_sample_threshold(dense, m.sample_stride, m.sample_stride, n_candidates, tg, n_active, stream);
    if (et != hipSuccess) return et;
    // 3. full pass with the filter: n_blocks workgroups, each writing its own half-segments and counts
    //    (the kernel dewi_timing_read reports: algorithmic bytes = n_rows * dim * 2 per launch)
    timing_begin(stream);
    hipLaunchKernelGGL(k_filter, dim3(m.n_blocks), dim3(kMfmaThreads), lds_bytes, stream, E, n_rows, qg,
                       m.n_tiles, static_cast<int64_t>(1), static_cast<const float*>(tg), og,
                       static_cast<int64_t>(m.seg_cap), cg, n_active, thr_bias);
    timing_end(stream);
  }
  return hipGetLastError();
}

hipError_t launch_prepare_queries_bf16(const float* d_Q, uint16_t* d_out, int n_queries, int n_rows_out, int dim, int space,
                                       float* d_qn2, hipStream_t stream) {
  hipLaunchKernelGGL(prepare_queries_bf16, dim3(n_rows_out), dim3(kWave), 0, stream, d_Q, d_out, n_queries, dim, space, d_qn2, 0);
  return hipGetLastError();
}

hipError_t launch_mfma_bf16(const MfmaLayout& m, const uint16_t* d_E, int64_t n_rows, int dim, const float* d_Q,
                            int n_queries, int n_candidates, int space, char* ws, int compute_units,
                            hipStream_t stream, float thr_bias) {
  // one launch per group of 256 would index rows from 0: the image of group g starts at g * 256 * dim, and the
  // fragment order is relative to the group's first query — (q / 32) etc. of the GLOBAL row index differ from the
  // group-local ones only by whole groups of 256 rows = 256 * dim elements, which is exactly the group's offset
  hipLaunchKernelGGL(prepare_queries_bf16, dim3(m.q_pad), dim3(kWave), 0, stream, d_Q,
                     reinterpret_cast<uint16_t*>(ws + m.qb_off), n_queries, dim, space, static_cast<float*>(nullptr), 1);
  switch (dim / 16) {
    case 8: return run_mfma_dim<8>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    case 16: return run_mfma_dim<16>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    case 24: return run_mfma_dim<24>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    case 32: return run_mfma_dim<32>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    case 40: return run_mfma_dim<40>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    case 48: return run_mfma_dim<48>(m, d_E, n_rows, n_queries, n_candidates, ws, compute_units, stream, thr_bias);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dewi
