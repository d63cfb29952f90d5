// Column-sliced SpMM for graphs whose node count fits the LDS: Y = (A with dropped edges) X, ACC += Y.
//
// The edge-parallel SpMM of lightgcn.hip gathers one 256-B source row per edge out of L2: nnz x D x 4 B = 381 MB
// per pass at the ML-1M graph (1.49 M edges, D 64), delivered at ~9.3 TB/s whatever the node order
// (profiles/r02_experiments.md §28) -- the kernel is bound by the L2 -> CU row-gather rate, 41 us per pass.
// The source matrix itself is tiny (9746 x 64 fp32 = 2.5 MB).  Cut into column slices of W = 4 floats it becomes
// 16 slices of N x 16 B = 156 KB: ONE slice fits a CU's 160 KB of LDS.  So the feature dimension is split over the
// workgroups instead of the lanes: block (slice s, row group g) loads slice s of X into LDS once (contiguous: the
// propagation buffers live in a SLICED layout [D/W][N][W]), then walks the edges of its rows reading sources from
// LDS.  What still comes out of L2 per edge is its (col, val) -- 6 B with 16-bit columns -- once per slice, and
// the 16 blocks that share a row group sit on the same XCD (blockIdx = s * n_groups + g, n_groups a multiple of
// 8), so the edge arrays leave HBM / MALL once.
//
// The graph is stored for this kernel (hiprec_sliced_csr): every row's edge list padded to a multiple of 16 slots
// (col 0, value 0), and cut into CHUNKS of at most 64 slots of one row.  A chunk is the work item of FOUR lanes,
// 16 consecutive slots each: three 16-byte loads per lane for the values, two for the columns, 16 LDS reads, 64
// FMAs, then two DPP steps over the quad and one ds_add_f32 per component into the row's accumulator in LDS (the
// rows of a block are cut into subgroups whose accumulators fit next to the slice).  Chunks are uniform, so their
// assignment is static (no row-length imbalance), descriptors are fetched two chunks ahead and edge data one chunk
// ahead of the arithmetic.  Edge dropout is applied to the value array once per step (drop_values_kernel), not
// per pass.  A subgroup's finished rows are written once, coalesced, with plain stores: no global atomics, no
// zero fill, and the fused layer sum is a plain read-modify-write.
//
// FACTORED graphs.  At 27 us per pass the kernel moves 16 x 9.4 MB of edge stream + 40 MB of slice fills out of L2,
// ~9.5 TB/s -- the rate the gather SpMM reached too: the L2 -> CU path is the bound, and only fewer bytes help.  A
// degree-normalised adjacency has rank-one values on its pattern, val[i][j] = row_scale[i] * col_scale[j]
// (lightgcn.py: factor_edge_values): the kernel then takes col_scale (.) X as its source, adds source rows without
// multiplying, applies row_scale[i] when a row is written, and hands col_scale (.) Y to the next pass -- the edge
// stream is the 2-byte column alone, and a dropped edge (or a padding slot) is a column that points at an all-zero
// row n of the slice.
//
// Earlier versions (MI355X, ML-1M graph, per pass): 16 lanes per ROW with a dynamic row counter 200 us (every row
// switch is a chain of dependent loads and the four groups of a wave serialise theirs); 16 lanes per 64-edge chunk,
// strided scalar loads, 16 DPP adds per chunk 47 us (VALU: ~180 instructions per 512 edges).
#include <algorithm>

#include "common.hpp"
#include "spmm.hpp"

namespace hiprec {

// Timing experiments (tools/exp_spmm_parts.py, profiles/r03_experiments.md 39): parts of the kernel switched off by the
// bits of HIPREC_SLICED_EXP -- 1 slice-major block map, 4 no output stores, 16 no slice fill, 32 no chunks at all
// (results are wrong with 4 / 16 / 32).  Only in builds made with -DHIPREC_SLICED_DEBUG (tools/build_debug_lib.sh); the product library has no
// such switch.
#ifdef HIPREC_SLICED_DEBUG
#define HIPREC_SLICED_EXP(bit) ((fl.exp & (bit)) != 0)
#else
#define HIPREC_SLICED_EXP(bit) false
#endif

// In-kernel timestamps (same builds): a first-wave and a last-wave thread of two workgroups note the cycle counter at the
// phase boundaries of the last launch; hiprec_debug_sliced_stamps reads them back (tools/exp_sliced_stamps.py).
#ifdef HIPREC_SLICED_DEBUG
__device__ unsigned long long g_sliced_stamps[4][16];
#define SLICED_STAMP(k)                                                                                          \
  do {                                                                                                           \
    if ((threadIdx.x == 0 || threadIdx.x == 960) && (blockIdx.x == 0 || blockIdx.x == 131) && (k) < 16)          \
      g_sliced_stamps[(blockIdx.x == 0 ? 0 : 2) + (threadIdx.x == 0 ? 0 : 1)][k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SLICED_STAMP(k) do {} while (0)
#endif

// 12 waves per workgroup (one workgroup per CU: the slice fills its LDS): three per SIMD keep the LDS pipe as busy as
// four did and a pass has a quarter fewer waves to launch, to meet at its barriers and to retire.  Same box, us per
// LightGCN / NGCF step: 1024 threads 115.3 / 216.9, 768: 113.9 / 215.8, 512: 118.3 (r06 experiments 79).  Any
// multiple of 64: a wave's 16 quads take one 16-chunk window per trip whatever the count.
constexpr int kSlicedThreads = 768;
constexpr int kSlicedQuads = kSlicedThreads / 4;
constexpr int64_t kSlicedLds = 160 * 1024 - 64;  // one workgroup's LDS, less the kernel's own few words
constexpr int kSlicedMinRowCap = 128;            // accumulator rows a subgroup must at least be able to hold
constexpr int kSlicedFill = (kSlicedLds / 16 + kSlicedThreads - 1) / kSlicedThreads;  // float4 per thread of a slice
constexpr int kSlicedMaxRowCap = 512;            // ... and at most: 2 output floats per thread at W = 4

template <int W>
struct SlicedVec;
template <>
struct SlicedVec<4> { using type = float __attribute__((ext_vector_type(4))); };
template <>
struct SlicedVec<2> { using type = float __attribute__((ext_vector_type(2))); };

// The S slots of one lane: S / 8 16-byte loads of 2-byte columns, and for a graph that keeps its values S / 4 more.
template <bool FACTORED, int S>
struct SlicedEdges {
  uint4 c[S / 8];
  float4 v[S / 4];
};
template <int S>
struct SlicedEdges<true, S> {
  uint4 c[S / 8];
};

// `edges`: float values [n_slots] (general graph) or uint16 columns [n_slots] (factored graph).  A lane without slots
// of its own (a chunk of fewer than 4 lanes' worth, or no chunk at all) reads the graph's all-padding tail instead of
// branching around its loads: every load of the loop is unconditional.
template <bool FACTORED, int S>
__device__ __forceinline__ SlicedEdges<FACTORED, S> load_sliced_edges(const uint16_t* __restrict__ col16,
                                                                      const void* __restrict__ edges, int2 d, int q,
                                                                      int pad_slot) {
  SlicedEdges<FACTORED, S> e;
  const bool live = q * S < ((d.y >> 16) & 0xFF);
  const int base = live ? (d.x & 0x3FFFFF) + q * S : pad_slot;  // a multiple of 8 slots: 16-B / 32-B aligned
  const uint4* pc = reinterpret_cast<const uint4*>((FACTORED ? static_cast<const uint16_t*>(edges) : col16) + base);
#pragma unroll
  for (int k = 0; k < S / 8; ++k) e.c[k] = pc[k];
  if constexpr (!FACTORED) {
    const float4* pv = reinterpret_cast<const float4*>(static_cast<const float*>(edges) + base);
#pragma unroll
    for (int k = 0; k < S / 4; ++k) e.v[k] = pv[k];
  }
  return e;
}

// acc_mode: 0 = none, 1 = accs += y, 2 = accs = y
// S = slots per lane and trip (hiprec_sliced_csr.lane_slots): a chunk is 4 S slots of one row.  What a trip does besides
// its S LDS reads per lane -- descriptor and edge prefetch, the quad's reduction, the run logic, the store -- costs about
// as many VALU instructions as 16 slots' address arithmetic and additions do, and at S = 16 the kernel was bound by the
// VALU (round 6, SQ counters: 59 % VALU busy, LDS 49 %): the host picks S per graph so that typical rows are few chunks.
//
// Where a finished row goes (round 6).  Rounds 2-5 summed every row in an LDS accumulator and wrote the accumulators
// of a SUBGROUP of rows in a flush phase between two workgroup barriers; in-kernel timestamps showed two subgroups'
// barriers and flushes plus the waves' wait for the slowest one taking 7 k of a pass's 31 k cycles.  Now the last quad
// of a run that IS its whole row (every row of at most 16 chunks that a window boundary does not cut) writes the row's
// four values itself, straight to global memory, with the row's factors and the layer sum's old values requested a
// trip ahead: a window of one-chunk rows stores 256 contiguous bytes.  Only rows that are several runs -- cut by a
// window boundary, or longer than a window -- still meet in LDS accumulators (`spill` rows, listed per workgroup by the
// host, index in the descriptor), flushed once at the end together with the rows that have no edges at all.
template <int W, bool FACTORED, int S>
__global__ __launch_bounds__(kSlicedThreads) void spmm_sliced_kernel(hiprec_sliced_csr a,
                                                                     const void* __restrict__ edges, float scale,
                                                                     const float* __restrict__ xs,
                                                                     float* __restrict__ ys, float* __restrict__ accs,
                                                                     int acc_mode, SlicedFlush fl, int dim) {
  static_assert(S % 8 == 0 && S >= 16 && 4 * S < 256, "slots per lane");
  SLICED_STAMP(0);
  float* __restrict__ zero_out = fl.zero_out;
  float* __restrict__ final_out = fl.final_out;
  const int final_set = fl.final_set ? 1 : 0;
  using Vec = typename SlicedVec<W>::type;
  using LdsVec = const __attribute__((address_space(3))) Vec;
  using Pair = float __attribute__((ext_vector_type(2)));
  using Edges = SlicedEdges<FACTORED, S>;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  const int64_t n_rows = a.n_rows;
  float* s_x = s_mem;                     // [n_rows + 1][W]: slice s of the source, then an all-zero row
  float* s_y = s_mem + (n_rows + 1) * W;  // [row_cap][W]: accumulators of the workgroup's spill rows
  const uint32_t row_bytes = W * sizeof(float);
  const uint32_t lds_base =
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)s_x));
  const int n_slices = static_cast<int>(gridDim.x) / a.n_groups;
  const int s = HIPREC_SLICED_EXP(1) ? static_cast<int>(blockIdx.x) % n_slices : static_cast<int>(blockIdx.x) / a.n_groups;
  const int g = HIPREC_SLICED_EXP(1) ? static_cast<int>(blockIdx.x) / n_slices : static_cast<int>(blockIdx.x) % a.n_groups;
  const int64_t slice_off = static_cast<int64_t>(s) * n_rows * W;
  const int quad = static_cast<int>(threadIdx.x) >> 2, q = static_cast<int>(threadIdx.x) & 3;
  const int lane = static_cast<int>(threadIdx.x) & 63;
  const int pad_slot = a.pad_slot;
  const int2* __restrict__ chunks = reinterpret_cast<const int2*>(a.chunks);
  const int c_begin = a.sub_chunk[g], c_end = a.sub_chunk[g + 1];  // the workgroup's chunks
  const int sp0 = a.spill_ptr[g], n_spill = a.spill_ptr[g + 1] - sp0;
  const int em0 = a.empty_ptr[g], n_empty = a.empty_ptr[g + 1] - em0;
  auto desc = [&](int c) { return c < c_end ? chunks[c] : int2{0, 0}; };  // {first slot | spill << 22, row | n_slots << 16 | run flags}
  // Quad k takes chunks k, k + 256, ... of the workgroup, with its next chunk's edge data and the descriptor after
  // that in flight ahead of the arithmetic.
  // the slice: every thread's (at most kSlicedFill) 16-byte loads are issued together -- a load-store loop would pay
  // the memory latency once per trip
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(xs + slice_off);
  const int n4 = static_cast<int>((n_rows * W) >> 2), n_tail = static_cast<int>((n_rows * W) & 3);  // (W = 2, odd n)
  float4 fill[kSlicedFill];
#pragma unroll
  for (int k = 0; k < kSlicedFill; ++k) {
    const int i = static_cast<int>(threadIdx.x) + k * kSlicedThreads;
    fill[k] = (i < n4 && !HIPREC_SLICED_EXP(16)) ? x4[i] : float4{0.f, 0.f, 0.f, 0.f};
  }
  int c = c_begin + quad;
  // d0 / e0: the chunk about to be summed; d1: the one after it (its edge data is requested while e0 is summed)
  int2 d0 = desc(c), d1 = desc(c + kSlicedQuads);
  SLICED_STAMP(1);
#pragma unroll
  for (int k = 0; k < kSlicedFill; ++k) {
    const int i = static_cast<int>(threadIdx.x) + k * kSlicedThreads;
    if (i < n4) reinterpret_cast<float4*>(s_x)[i] = fill[k];
  }
  SLICED_STAMP(2);
  if (static_cast<int>(threadIdx.x) < n_tail) s_x[4 * n4 + threadIdx.x] = xs[slice_off + 4 * n4 + threadIdx.x];
  if (static_cast<int>(threadIdx.x) < W) s_x[n_rows * W + threadIdx.x] = 0.f;
  Edges e0 = load_sliced_edges<FACTORED, S>(a.col16, edges, d0, q, pad_slot);
  for (int i = threadIdx.x; i < n_spill * W; i += kSlicedThreads) s_y[i] = 0.f;
  __syncthreads();
  SLICED_STAMP(3);

  // What becomes of component q of row r's sum v: the layer output (times the factors of a factored graph), the layer
  // sum, or -- the last pass of a propagation -- the finished sum in the row-major result.
  struct RowIn {  // what emit needs of the row besides its sum, requested before the sum is there
    float old = 0.f, rs = 1.f, cs = 1.f, fin = 0.f;
  };
  auto row_in = [&](int r, int cq) {
    RowIn in;
    if (acc_mode == 1) in.old = accs[slice_off + static_cast<int64_t>(r) * W + cq];
    if constexpr (FACTORED) {
      in.rs = a.row_scale[r];
      in.cs = a.col_scale[r];
    }
    if (final_out != nullptr && !final_set) in.fin = final_out[static_cast<int64_t>(r) * dim + s * W + cq];
    return in;
  };
  auto emit = [&](int r, int cq, float v, const RowIn& in) {
    if (HIPREC_SLICED_EXP(4) && v != 1e30f) return;
    float y = v * scale, y_next = y;
    if constexpr (FACTORED) {  // row factor now; the next pass wants its source scaled by the column factor
      y *= in.rs;
      y_next = y * in.cs;
    }
    const int64_t o = slice_off + static_cast<int64_t>(r) * W + cq;
    if (final_out != nullptr) {  // the result (plus the layer sum so far, acc_mode 1) goes straight to row-major
      float* dst = final_out + static_cast<int64_t>(r) * dim + s * W + cq;
      *dst = final_set ? in.old + y : in.fin + (in.old + y);
    } else {
      ys[o] = y_next;
      if (acc_mode == 1) accs[o] = in.old + y;
      else if (acc_mode == 2) accs[o] = y;
    }
    if (zero_out != nullptr) zero_out[o] = 0.f;
  };

  // One chunk: S random source rows out of the LDS per lane (~1.3-way bank conflicts after the host's slot
  // permutation), summed; the quad's four partial sums folded so that lane q holds component q; the wave's runs of
  // chunks of one row summed; the last quad of a run stores.
  auto sum_chunk = [&](const Edges& e, const int2 d) {
    const uint32_t dy = static_cast<uint32_t>(d.y);
    const int row = static_cast<int>(dy & 0xFFFF);
    const bool ends = q < W && ((dy >> 27) & 1), direct = ends && ((dy >> 28) & 1);
    RowIn in;
    if (direct) in = row_in(row, q);  // (the loads return while the slots are summed)
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(e.c);  // two columns per word
    Vec src[2][4];
    Pair acc[W / 2];  // two floats per register pair: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
    for (int h = 0; h < W / 2; ++h) acc[h] = Pair{0.f, 0.f};
    // LDS byte address of slot j's source row: column (low / high half of a word) * row bytes + base, one
    // v_mad_u32_u16 each; four reads are requested at a time, eight are in flight while the previous four are added --
    // left alone the compiler keeps two reads in flight and waits for each
    auto request = [&](int grp) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int slot = 4 * grp + j;
        uint32_t addr;
        if (slot & 1)
          asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(cw[slot >> 1]), "v"(row_bytes), "v"(lds_base));
        else
          asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(addr) : "v"(cw[slot >> 1]), "v"(row_bytes), "v"(lds_base));
        src[grp & 1][j] = *reinterpret_cast<LdsVec*>(addr);
      }
    };
    __builtin_amdgcn_sched_barrier(0);
    request(0);
#pragma unroll
    for (int b = 0; b < S / 4; ++b) {
      if (b + 1 < S / 4) request(b + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Pair* f = reinterpret_cast<const Pair*>(&src[b & 1][j]);
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {
          if constexpr (FACTORED) {
            acc[h] += f[h];
          } else {
            const float* vv = reinterpret_cast<const float*>(e.v);
            acc[h] += Pair{vv[4 * b + j], vv[4 * b + j]} * f[h];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float mine = 0.f;  // lane q < W of the quad adds component q
#pragma unroll
    for (int w = 0; w < W; ++w) {
      float t = dpp_add<0xB1>(acc[w >> 1][w & 1]);  // quad_perm [1,0,3,2]
      t = dpp_add<0x4E>(t);             // quad_perm [2,3,0,1]
      mine = q == w ? t : mine;
    }
    // A row's chunks are consecutive, so the quads of a wave that work on one row are neighbours -- and LDS float
    // atomics are slow (~3 cycles per LANE, profiles/r03_experiments.md 39).  So the wave sums a row's quads itself:
    // a segmented inclusive scan over its 16 quads -- two DPP steps inside every 16-lane row, then the total so far
    // carried from row to row through SGPRs -- and the LAST quad of a run stores.  Which quad does what is static (a
    // wave's 16 quads always hold one window of 16 chunks) and comes with the descriptor: lightgcn.py
    // _windowed_chunks.  A window of one-chunk rows (the common one once S fits the typical row) skips all of it.
    if (__ballot(((dy >> 24) & 7) != 0) != 0) {
      const int in_row = (dy >> 24) & 3;
      const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0x114, 0xF, 0xF, false));  // row_shr:4
      if (in_row >= 1) mine += o1;
      const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0x118, 0xF, 0xF, false));  // row_shr:8
      if (in_row >= 2) mine += o2;
#pragma unroll
      for (int r = 1; r < 4; ++r) {  // in order: row r - 1's last quad has its own carry by now
        const bool need = ((dy >> 26) & 1) && (lane >> 4) == r;
        if (__ballot(need) != 0) {
          const int mi = __builtin_bit_cast(int, mine);
          const int k0 = __builtin_amdgcn_readlane(mi, 16 * r - 4), k1 = __builtin_amdgcn_readlane(mi, 16 * r - 3),
                    k2 = __builtin_amdgcn_readlane(mi, 16 * r - 2), k3 = __builtin_amdgcn_readlane(mi, 16 * r - 1);
          if (need) mine += __builtin_bit_cast(float, q == 0 ? k0 : q == 1 ? k1 : q == 2 ? k2 : k3);
        }
      }
    }
    if (direct) emit(row, q, mine, in);
    else if (ends) lds_add_f32(&s_y[((static_cast<uint32_t>(d.x) >> 22) & 0x1FF) * W + q], mine);
  };

  // two trips per turn of the loop, the edge registers of one filled while the other's are summed: written as a
  // one-deep rotation the compiler copied every edge register once per trip (25 v_mov of a 16-slot trip's 165)
  if (HIPREC_SLICED_EXP(32)) c = c_end + quad;
  SLICED_STAMP(4);
  while (c < c_end) {
    const int2 d2 = desc(c + 2 * kSlicedQuads);
    const Edges e1 = load_sliced_edges<FACTORED, S>(a.col16, edges, d1, q, pad_slot);
    sum_chunk(e0, d0);
    c += kSlicedQuads;
    if (!(c < c_end)) break;
    const int2 d3 = desc(c + 2 * kSlicedQuads);
    e0 = load_sliced_edges<FACTORED, S>(a.col16, edges, d2, q, pad_slot);
    sum_chunk(e1, d1);
    c += kSlicedQuads;
    d0 = d2;
    d1 = d3;
  }
  SLICED_STAMP(5);
  // the rows that are not one run, and the rows without edges: threads 0 .. (n_spill + n_empty) W take one value each
  const int n_late = (n_spill + n_empty) * W;
  if (n_late == 0) {
    SLICED_STAMP(15);
    return;
  }
  __syncthreads();
  SLICED_STAMP(6);
  for (int i = threadIdx.x; i < n_late; i += kSlicedThreads) {
    const int k = W == 4 ? i >> 2 : i >> 1, cq = i & (W - 1);
    const int r = k < n_spill ? a.spill_row[sp0 + k] : a.empty_row[em0 + k - n_spill];
    const RowIn in = row_in(r, cq);
    emit(r, cq, k < n_spill ? s_y[i] : 0.f, in);
  }
  SLICED_STAMP(15);
}

// out[slot] = keep[eid[slot]] ? val[slot] : 0 (padding slots: eid < 0, value 0) -- the dropped edge values of a
// step in the sliced graph's slot order, so that no pass looks at keep bytes; for a factored graph the step's
// stream is uint16 columns instead, col16[slot] or n_rows (the zero row).  One launch serves both graphs of a
// plan (slots of `a`, then slots of `b`); with `draw` the keep decision is the counter-based device draw itself
// (the same function of (seed, step, edge) in both graphs) and `a`'s slots publish it to keep[] when that is given.
// OPT >= 0 (round 4): the launch is ALSO the dense optimizer step of the matrix it lays out -- in.x is the parameter
// buffer w, `fuse` names g / m / v: a thread updates its W floats (the shared opt_update, block 0 folds the loss
// partials like opt_dense_kernel) and writes the fresh values row-major AND sliced.  The optimizer launch of step t then
// prepares step t + 1 (edge streams drawn for `step`, E0 laid out): one launch instead of two, no re-read of w.
template <int OPT>
__global__ __launch_bounds__(kBlock) void step_values_kernel(hiprec_sliced_csr a, hiprec_sliced_csr b,
                                                             uint8_t* __restrict__ keep, int draw, float keep_prob,
                                                             uint64_t seed, uint64_t step, float* __restrict__ out_a,
                                                             float* __restrict__ out_b, SlicedInput in, SlicedOpt fuse) {
  float step_size = 0.f, bc2_sqrt = 1.f;
  if constexpr (OPT >= 0) step_scalars<OPT>(fuse.s, fuse.stats, &step_size, &bc2_sqrt);
  // eight consecutive slots per thread (n_slots is a multiple of 16): 16-byte loads and stores
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock, total = (a.n_slots + b.n_slots) >> 3;
  // ... and, riding on the same launch (its FIRST threads: not a tail), the step's input matrix into the sliced layout
  // (to_sliced_kernel's work): one W-float group per thread, 32-bit index arithmetic
  const int per_row = in.x != nullptr ? in.dim / in.W : 0;
  const int64_t n_in = static_cast<int64_t>(per_row) * in.n_rows;  // < 2^31: n_rows < 65536
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total + n_in; i += stride) {
    if (i < n_in) {
      const uint32_t j = static_cast<uint32_t>(i), r = j / static_cast<uint32_t>(per_row),
                     sl = j - r * static_cast<uint32_t>(per_row);
      const float* src = in.x + static_cast<int64_t>(r) * in.dim + sl * in.W;
      const int64_t o = (static_cast<int64_t>(sl) * in.n_rows + r) * in.W;
      const float f = in.row_scale ? in.row_scale[r] : 1.f;
      if (in.W == 4) {
        float4 v = *reinterpret_cast<const float4*>(src);
        if constexpr (OPT >= 0) {
          const int64_t at = static_cast<int64_t>(r) * in.dim + sl * in.W;
          float4 gv = *reinterpret_cast<const float4*>(fuse.g + at), mv = float4{0.f, 0.f, 0.f, 0.f},
                 vv = float4{0.f, 0.f, 0.f, 0.f};
          if constexpr (OPT == HIPREC_OPT_ADAM) mv = *reinterpret_cast<const float4*>(fuse.m + at);
          if constexpr (OPT != HIPREC_OPT_SGD) vv = *reinterpret_cast<const float4*>(fuse.v + at);
          opt_update<OPT>(v.x, gv.x, mv.x, vv.x, fuse.s, step_size, bc2_sqrt);
          opt_update<OPT>(v.y, gv.y, mv.y, vv.y, fuse.s, step_size, bc2_sqrt);
          opt_update<OPT>(v.z, gv.z, mv.z, vv.z, fuse.s, step_size, bc2_sqrt);
          opt_update<OPT>(v.w, gv.w, mv.w, vv.w, fuse.s, step_size, bc2_sqrt);
          *reinterpret_cast<float4*>(fuse.w + at) = v;
          *reinterpret_cast<float4*>(fuse.g + at) = gv;
          if constexpr (OPT == HIPREC_OPT_ADAM) *reinterpret_cast<float4*>(fuse.m + at) = mv;
          if constexpr (OPT != HIPREC_OPT_SGD) *reinterpret_cast<float4*>(fuse.v + at) = vv;
        }
        *reinterpret_cast<float4*>(in.xs + o) = float4{v.x * f, v.y * f, v.z * f, v.w * f};
        if (in.xs_copy) *reinterpret_cast<float4*>(in.xs_copy + o) = v;
      } else {
        for (int c = 0; c < in.W; ++c) {
          in.xs[o + c] = src[c] * f;
          if (in.xs_copy) in.xs_copy[o + c] = src[c];
        }
      }
      continue;
    }
    const int64_t w8 = (i - n_in) << 3;
    const bool first = w8 < a.n_slots;
    const int64_t e = first ? w8 : w8 - a.n_slots;
    const hiprec_sliced_csr& gph = first ? a : b;
    const int4 k0 = *reinterpret_cast<const int4*>(gph.eid + e), k1 = *reinterpret_cast<const int4*>(gph.eid + e + 4);
    const int32_t k[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    bool kept[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      kept[j] = false;
      if (k[j] >= 0) {
        if (draw) {
          kept[j] = keep_draw(seed, step, k[j], keep_prob);
          if (first && keep != nullptr) keep[k[j]] = kept[j] ? 1 : 0;
        } else {
          kept[j] = keep[k[j]] != 0;
        }
      }
    }
    float* out = first ? out_a : out_b;
    if (gph.col_scale != nullptr) {  // factored: a dropped edge points at the zero row
      const uint4 c = *reinterpret_cast<const uint4*>(gph.col16 + e);
      const uint32_t cw[4] = {c.x, c.y, c.z, c.w}, z = static_cast<uint32_t>(gph.n_rows);
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (kept[2 * j] ? cw[j] & 0xFFFFu : z) | (kept[2 * j + 1] ? cw[j] & 0xFFFF0000u : z << 16);
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + e) = uint4{o[0], o[1], o[2], o[3]};
    } else {
      const float4 v0 = *reinterpret_cast<const float4*>(gph.val + e), v1 = *reinterpret_cast<const float4*>(gph.val + e + 4);
      *reinterpret_cast<float4*>(out + e) =
          float4{kept[0] ? v0.x : 0.f, kept[1] ? v0.y : 0.f, kept[2] ? v0.z : 0.f, kept[3] ? v0.w : 0.f};
      *reinterpret_cast<float4*>(out + e + 4) =
          float4{kept[4] ? v1.x : 0.f, kept[5] ? v1.y : 0.f, kept[6] ? v1.z : 0.f, kept[7] ? v1.w : 0.f};
    }
  }
  if constexpr (OPT >= 0) {
    if (blockIdx.x == 0 && fuse.scratch) finalize_partials(fuse.stats, fuse.scratch);
  }
}

// row-major [n_rows][dim]  <->  sliced [dim / W][n_rows][W]
// (xs = row_scale (.) x when row_scale is given: the source of a factored graph's pass; xs_copy = x)
__global__ __launch_bounds__(kBlock) void to_sliced_kernel(const float* __restrict__ x, int64_t n_rows, int dim,
                                                           int W, const float* __restrict__ row_scale,
                                                           float* __restrict__ xs, float* __restrict__ xs_copy) {
  const int64_t total = n_rows * dim;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / dim;
    const int c = static_cast<int>(i - r * dim);
    const int64_t o = (static_cast<int64_t>(c / W) * n_rows + r) * W + c % W;
    const float v = x[i];
    xs[o] = row_scale ? v * row_scale[r] : v;
    if (xs_copy) xs_copy[o] = v;
  }
}

// y (row-major) = or += xs (sliced)
__global__ __launch_bounds__(kBlock) void from_sliced_kernel(const float* __restrict__ xs, int64_t n_rows, int dim,
                                                             int W, float* __restrict__ y, int add) {
  const int64_t total = n_rows * dim;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / dim;
    const int c = static_cast<int>(i - r * dim);
    const float v = xs[(static_cast<int64_t>(c / W) * n_rows + r) * W + c % W];
    y[i] = add ? y[i] + v : v;
  }
}

// LDS one workgroup may use on the current device (gfx950: 160 KB); a part with less (gfx942: 64 KB) gets width 0
// from sliced_width for graphs whose slice does not fit, i.e. the edge-parallel gather SpMM
static int64_t device_lds_bytes() {
  static int64_t cached = -1;
  if (cached < 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
      cached = v;
    else
      cached = 0;   // no device (the CPU-side symbol tests): keep the gfx950 figure below
  }
  return cached;
}

int sliced_width(int64_t n_rows, int dim) {
  if (n_rows <= 0 || n_rows >= 65536 || dim <= 0) return 0;  // 16-bit column and row ids
  const int64_t have = device_lds_bytes();
  if (have > 0 && have < kSlicedLds) return 0;
  // a width below 4 floats re-reads the edge arrays more often than the gather SpMM reads source rows: W = 4 or
  // (a narrower slice for graphs of up to ~20 k nodes) 2
  for (int w : {4, 2})
    if (dim % w == 0 && (n_rows + 1 + kSlicedMinRowCap) * w * static_cast<int64_t>(sizeof(float)) <= kSlicedLds)
      return w;
  return 0;
}

int sliced_row_cap(int64_t n_rows, int dim) {
  const int w = sliced_width(n_rows, dim);
  if (w == 0) return 0;
  return static_cast<int>(std::min<int64_t>(kSlicedLds / (w * sizeof(float)) - n_rows - 1, kSlicedMaxRowCap));
}

template <int W, bool FACTORED, int S>
static int launch_sliced_as(const hiprec_sliced_csr* a, const void* edges, float scale, const float* xs, float* ys,
                            float* accs, int acc_mode, int dim, size_t lds, hipStream_t st, const SlicedFlush& fl) {
  static std::atomic<uint64_t> lds_ok{0};  // per instantiation: the limit is an attribute of the function
  if (int rc = allow_dynamic_lds({reinterpret_cast<const void*>(&spmm_sliced_kernel<W, FACTORED, S>)}, kSlicedLds, lds_ok,
                                 "the column-sliced SpMM"))
    return rc;
  const int grid = (dim / W) * a->n_groups;
  spmm_sliced_kernel<W, FACTORED, S><<<grid, kSlicedThreads, lds, st>>>(*a, edges, scale, xs, ys, accs, acc_mode, fl, dim);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

int launch_spmm_sliced(const hiprec_sliced_csr* a, const void* edges, float scale, const float* xs, float* ys,
                       float* accs, int acc_mode, int dim, int W, hipStream_t st, SlicedFlush fl) {
  HIPREC_REQUIRE(a && a->sub_row && a->sub_chunk && a->n_rows > 0 && a->n_groups > 0 && a->subs_per_group == 1,
                 "bad sliced graph (one range of chunks per workgroup: subs_per_group == 1)");
  HIPREC_REQUIRE(a->spill_ptr && a->empty_ptr && a->spill_row && a->empty_row, "sliced graph without spill / empty row lists");
  HIPREC_REQUIRE(a->n_slots < (1ll << 22), "%lld slots: the descriptors hold 22 bits of first slot", (long long)a->n_slots);
  HIPREC_REQUIRE(a->n_slots > 0 && a->col16 && a->val && (a->n_chunks == 0 || a->chunks),
                 "sliced graph has NULL chunks / col16 / val");
  HIPREC_REQUIRE(a->n_slots % 16 == 0, "n_slots %lld is not a multiple of 16", (long long)a->n_slots);
  HIPREC_REQUIRE(a->n_chunks % 16 == 0,
                 "%d chunks: the descriptors must be laid out in windows of 16 with their run flags (include/hiprec.h, "
                 "hiprec_sliced_csr; lightgcn._windowed_chunks builds them)", a->n_chunks);
  HIPREC_REQUIRE((a->row_scale == nullptr) == (a->col_scale == nullptr), "row_scale and col_scale go together");
  const bool factored = a->col_scale != nullptr;
  const int S = a->lane_slots;
  HIPREC_REQUIRE(S == 16 || (factored && (S == 24 || S == 32 || S == 48)),
                 "lane_slots %d: 16, or 24 / 32 / 48 for a factored graph", S);
  HIPREC_REQUIRE(a->pad_slot >= 0 && a->pad_slot % 8 == 0 && a->pad_slot + S <= a->n_slots,
                 "pad_slot %d: the graph must end with lane_slots = %d padding slots (n_slots %lld)", a->pad_slot, S,
                 (long long)a->n_slots);
  HIPREC_REQUIRE(W > 0 && W == sliced_width(a->n_rows, dim), "slice width %d does not fit %lld rows x dim %d", W,
                 (long long)a->n_rows, dim);
  HIPREC_REQUIRE(a->row_cap > 0 && a->row_cap <= sliced_row_cap(a->n_rows, dim),
                 "subgroups of up to %d rows do not fit the LDS next to the slice (at most %d)", a->row_cap,
                 sliced_row_cap(a->n_rows, dim));
  HIPREC_REQUIRE(xs && (ys || fl.final_out) && (acc_mode == 0 || accs), "NULL sliced buffers");
  const size_t lds = static_cast<size_t>(a->n_rows + 1 + a->row_cap) * W * sizeof(float);
  if (edges == nullptr) edges = factored ? static_cast<const void*>(a->col16) : static_cast<const void*>(a->val);
#ifdef HIPREC_SLICED_DEBUG
  if (const char* e = getenv("HIPREC_SLICED_EXP")) fl.exp = atoi(e);
#endif
#define HIPREC_SLICED_CASE(w, f, sl) \
  if (W == w && factored == f && S == sl) \
    return launch_sliced_as<w, f, sl>(a, edges, scale, xs, ys, accs, acc_mode, dim, lds, st, fl);
  HIPREC_SLICED_CASE(4, true, 16)
  HIPREC_SLICED_CASE(4, true, 24)
  HIPREC_SLICED_CASE(4, true, 32)
  HIPREC_SLICED_CASE(4, true, 48)
  HIPREC_SLICED_CASE(2, true, 16)
  HIPREC_SLICED_CASE(2, true, 24)
  HIPREC_SLICED_CASE(2, true, 32)
  HIPREC_SLICED_CASE(2, true, 48)
  HIPREC_SLICED_CASE(4, false, 16)
  HIPREC_SLICED_CASE(2, false, 16)
#undef HIPREC_SLICED_CASE
  set_error("no column-sliced SpMM for width %d, lane_slots %d", W, S);
  return HIPREC_E_UNSUPPORTED;
}

int launch_step_values(const hiprec_sliced_csr* a, const hiprec_sliced_csr* b, uint8_t* keep, bool draw,
                       float keep_prob, uint64_t seed, uint64_t step, float* out_a, float* out_b, hipStream_t st,
                       SlicedInput in, const SlicedOpt* fuse) {
  hiprec_sliced_csr none = {};
  if (b == nullptr) b = &none;
  HIPREC_REQUIRE(a && out_a && (a->n_slots == 0 || (a->val && a->eid)), "bad sliced graph");
  HIPREC_REQUIRE(b->n_slots == 0 || (b->val && b->eid && out_b), "bad second sliced graph");
  HIPREC_REQUIRE(draw || keep, "keep bytes needed");
  HIPREC_REQUIRE(in.x == nullptr || (in.W > 0 && in.dim % in.W == 0 && in.n_rows < 65536), "bad sliced input");
  const int64_t n_in = in.x != nullptr ? in.n_rows * (in.dim / in.W) : 0;
  if (a->n_slots + b->n_slots + n_in == 0) return 0;
  HIPREC_REQUIRE(a->n_slots % 16 == 0 && b->n_slots % 16 == 0, "n_slots is not a multiple of 16");
  const int grid = grid_for_threads((a->n_slots + b->n_slots) / 8 + n_in);
  const int dr = draw ? 1 : 0;
  if (fuse == nullptr) {
    step_values_kernel<-1><<<grid, kBlock, 0, st>>>(*a, *b, keep, dr, keep_prob, seed, step, out_a, out_b, in, SlicedOpt{});
  } else {
    HIPREC_REQUIRE(in.x != nullptr && in.W == 4 && in.x == fuse->w && fuse->g && fuse->stats,
                   "the fused optimizer needs the parameter buffer as the sliced input, slice width 4");
    switch (fuse->kind) {
      case HIPREC_OPT_SGD:
        step_values_kernel<HIPREC_OPT_SGD><<<grid, kBlock, 0, st>>>(*a, *b, keep, dr, keep_prob, seed, step, out_a, out_b, in, *fuse);
        break;
      case HIPREC_OPT_ADAM:
        HIPREC_REQUIRE(fuse->m && fuse->v, "adam needs exp_avg / exp_avg_sq buffers");
        step_values_kernel<HIPREC_OPT_ADAM><<<grid, kBlock, 0, st>>>(*a, *b, keep, dr, keep_prob, seed, step, out_a, out_b, in, *fuse);
        break;
      case HIPREC_OPT_RMSPROP:
        HIPREC_REQUIRE(fuse->v, "rmsprop needs a square_avg buffer");
        step_values_kernel<HIPREC_OPT_RMSPROP><<<grid, kBlock, 0, st>>>(*a, *b, keep, dr, keep_prob, seed, step, out_a, out_b, in, *fuse);
        break;
      default:
        set_error("unknown optimizer kind %d", fuse->kind);
        return HIPREC_E_UNSUPPORTED;
    }
  }
  HIPREC_TRY(hipGetLastError());
  return 0;
}

int launch_to_sliced(const float* x, int64_t n_rows, int dim, int W, const float* row_scale, float* xs,
                     float* xs_copy, hipStream_t st) {
  to_sliced_kernel<<<grid_for_threads(n_rows * dim), kBlock, 0, st>>>(x, n_rows, dim, W, row_scale, xs, xs_copy);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

int launch_from_sliced(const float* xs, int64_t n_rows, int dim, int W, float* y, bool add, hipStream_t st) {
  from_sliced_kernel<<<grid_for_threads(n_rows * dim), kBlock, 0, st>>>(xs, n_rows, dim, W, y, add ? 1 : 0);
  HIPREC_TRY(hipGetLastError());
  return 0;
}

}  // namespace hiprec

using namespace hiprec;

#ifdef HIPREC_SLICED_DEBUG
extern "C" int hiprec_debug_sliced_stamps(unsigned long long* out64) {
  HIPREC_TRY(hipDeviceSynchronize());
  HIPREC_TRY(hipMemcpyFromSymbol(out64, HIP_SYMBOL(hiprec::g_sliced_stamps), sizeof(unsigned long long) * 64));
  return 0;
}
#endif

extern "C" int32_t hiprec_sliced_width(int64_t n_rows, int32_t dim) { return sliced_width(n_rows, dim); }

extern "C" int32_t hiprec_sliced_row_cap(int64_t n_rows, int32_t dim) { return sliced_row_cap(n_rows, dim); }

extern "C" int hiprec_to_sliced(const float* x, int64_t n_rows, int32_t dim, int32_t slice_w, const float* row_scale,
                                float* xs, void* stream) {
  HIPREC_REQUIRE(x && xs && n_rows > 0 && dim > 0 && slice_w > 0 && dim % slice_w == 0, "bad arguments");
  return launch_to_sliced(x, n_rows, dim, slice_w, row_scale, xs, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_from_sliced(const float* xs, int64_t n_rows, int32_t dim, int32_t slice_w, float* y,
                                  int32_t add, void* stream) {
  HIPREC_REQUIRE(y && xs && n_rows > 0 && dim > 0 && slice_w > 0 && dim % slice_w == 0, "bad arguments");
  return launch_from_sliced(xs, n_rows, dim, slice_w, y, add != 0, static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_sliced_drop_values(const hiprec_sliced_csr* a, const uint8_t* keep, float* out, void* stream) {
  return launch_step_values(a, nullptr, const_cast<uint8_t*>(keep), false, 1.f, 0, 0, out, nullptr,
                            static_cast<hipStream_t>(stream));
}

extern "C" int hiprec_spmm_sliced(const hiprec_sliced_csr* a, const void* step_edges, float scale, const float* xs,
                                  float* ys, float* accs, int32_t acc_mode, int32_t dim, int32_t slice_w,
                                  void* stream) {
  return launch_spmm_sliced(a, step_edges, scale, xs, ys, accs, acc_mode, dim, slice_w,
                            static_cast<hipStream_t>(stream), SlicedFlush{});
}
