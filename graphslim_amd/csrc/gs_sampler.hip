// Class-wise two-hop neighbour sampler, all classes in one launch pair.
// Replaces dataset/loader.py:187-224 (np.random.permutation seeds + PyG NeighborSampler ->
// torch_sparse.sample_adj on the CPU, followed by a per-class H2D copy, gcond_base.py:212-213).
//
// Kernel 1 (one workgroup per class): draw the seeds (Floyd's algorithm over the class pool), sample
// <= f1 neighbours per seed without replacement, de-duplicate the frontier with a global marker
// array (first occurrence in edge-slot order wins -> deterministic local ids, seeds first), and build
// the CSC of the hop-1 block (used by the transposed product in the real-side backward).
// Kernel 2 (one thread per frontier node): sample <= f2 neighbours; sources stay global ids, because
// the outermost hop only ever indexes X (or relu(XW1+b1)) rows — no relabel, no second de-dup.
// Randomness: counter-based hash keyed by (seed, step, class, row, draw): replayable, order-free.
#include "gs_common.h"

#define GS_MAX_BCAP 256
#define GS_MAX_F 16
#define GS_MARK_FREE 0x7fffffff
#define GS_MARK_BIAS 0x40000000

__device__ __forceinline__ int mk_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mk_store(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int block_excl_scan_256(int v, int* s_tmp, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) s_tmp[w] = x;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += s_tmp[i];
  *total = s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
  __syncthreads();
  return base + x - v;
}

// sample k = min(deg, f) distinct offsets in [0, deg) into sel[0..k); returns k.  sel lives in LDS.
__device__ __forceinline__ int sample_row(int deg, int f, int* sel, uint64_t seed, uint32_t a, uint32_t b) {
  if (deg <= f) {
    for (int j = 0; j < deg; ++j) sel[j] = j;
    return deg;
  }
  int ns = 0;
  for (int j = deg - f; j < deg; ++j) {   // Floyd: uniform f-subset of [0, deg)
    int t = (int)gs_rand_below(seed, a, b, (uint32_t)ns, (uint32_t)(j + 1));
    bool found = false;
    for (int q = 0; q < ns; ++q) found |= (sel[q] == t);
    sel[ns++] = found ? j : t;
  }
  return f;
}

__global__ __launch_bounds__(256) void gs_sample_hop1_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ val, int n_nodes,
                                                             const int* __restrict__ cls_ptr,
                                                             const int* __restrict__ cls_nodes, int* marker,
                                                             gs_blocks blk, uint64_t seed, uint32_t step, int class_key0,
                                                             int seeds_only, int cand_cap) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int ckey = class_key0 + c;   // RNG key = global class id (a class shard draws what the full run draws)
  const int bcap = blk.bcap, f1 = blk.f1, cap1 = blk.cap1;
  const int* pool = cls_nodes + cls_ptr[c];
  const int npool = cls_ptr[c + 1] - cls_ptr[c];
  int B = min(bcap, npool);
  int* mk = marker + (long)c * n_nodes;

  __shared__ int s_g[GS_MAX_BCAP * GS_MAX_F];          // global id per hop-1 edge slot (-1 = padding)
  __shared__ int s_sel[GS_MAX_BCAP * GS_MAX_F];        // per-thread Floyd scratch
  __shared__ int s_cnt[GS_MAX_BCAP * (GS_MAX_F + 1) + 1];
  __shared__ int s_seed[GS_MAX_BCAP];
  __shared__ int s_hash[1024];
  __shared__ int s_tmp[4];

  // ---- seeds: B distinct pool entries (loader.py:222: np.random.permutation(pool)[:num]) ----------
  if (npool <= bcap) {
    if (tid < B) s_seed[tid] = pool[tid];
  } else {
    // A uniform B-subset of the pool by ALL 256 threads (the first form drew it with Floyd's algorithm on thread 0: 256 dependent
    // steps, most of the kernel's time): every pool index i gets the key (hash(seed, step, class, i), i); the B smallest keys
    // are the batch — a uniform subset because the keys are i.i.d. and ties are broken by the index.  Selection without a
    // sort: histogram of the top 12 key bits, the bin T in which the B-th smallest falls, every index below T, and the
    // (need) smallest of bin T by rank among the (few) members of that bin.  Seeds keep ascending pool order.
    int* hist = s_g;                               // 4096 bins (s_g / s_sel / s_cnt are not in use yet)
    int* cand = s_sel;                             // members of the crossing bin: [2 m] = (key, index) pairs, m <= cand_cap <= 2048
    int* take = s_cnt;                             // take[a] = candidate a is among the `need` smallest (its own array: the rank pass
                                                   // only READS cand, so no wave can see a half-updated candidate list)
    const int per = (npool + 255) / 256, i_lo = min(tid * per, npool), i_hi = min(i_lo + per, npool);
    // level 0 bins the top 12 key bits; if the crossing bin holds more than cand_cap members (pools beyond ~ 4096 cand_cap), level 1
    // bins the next 12 bits of that bin's members, so that the ranked set stays small.  kmask / kpref select the members of the
    // bins fixed so far; `below` counts the indices already known to be taken (keys in front of the crossing bin).
    uint32_t kmask = 0u, kpref = 0u;
    int below = 0, shift = 20, mcount = 0;
    for (int level = 0; level < 2; ++level) {
      for (int i = tid; i < 4096; i += 256) hist[i] = 0;
      __syncthreads();
      for (int i = i_lo; i < i_hi; ++i) {
        const uint32_t key = gs_rand_u32(seed ^ 0x5eedull, step, (uint32_t)ckey, (uint32_t)i);
        if ((key & kmask) == kpref) atomicAdd(&hist[(key >> shift) & 4095u], 1);
      }
      __syncthreads();
      int hsum = 0;
      for (int q = 0; q < 16; ++q) hsum += hist[tid * 16 + q];
      int htot = 0;
      const int hbase = below + block_excl_scan_256(hsum, s_tmp, &htot);   // indices in front of this thread's 16 bins
      if (hbase < B && hbase + hsum >= B) {                                  // the crossing lies in this thread's bins (exactly one thread)
        int acc = hbase;
        for (int q = 0; q < 16; ++q) {
          const int hq = hist[tid * 16 + q];
          if (acc + hq >= B) { s_hash[0] = tid * 16 + q; s_hash[1] = acc; s_hash[2] = hq; break; }
          acc += hq;
        }
      }
      __syncthreads();
      kmask |= 4095u << shift;
      kpref |= (uint32_t)s_hash[0] << shift;
      below = s_hash[1];
      mcount = s_hash[2];
      __syncthreads();
      if (mcount <= cand_cap) break;
      shift -= 12;
    }
    const int need = B - below;
    if (tid == 0) s_hash[3] = 0;
    __syncthreads();
    // Each thread's slice of the pool is at most 128 indices for pools up to 32 768 (every BASELINE configuration): the class of each
    // index — in front of the crossing bin / inside it — is then kept as two 128-bit masks in registers, so that the count and write
    // passes below do not hash the pool again (the first form evaluated the 64-bit hash four times per index: 43 us of the 91 us the
    // 18 000-node class of the ogbn-arxiv shape took, round-3 phase stamps).  Larger slices recompute.
    const bool use_mask = per <= 128;
    uint32_t m_below[4] = {0u, 0u, 0u, 0u}, m_in[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      uint32_t bb = 0u, bi = 0u;
      for (int bit = 0; bit < 32; ++bit) {
        const int i = i_lo + 32 * w + bit;
        if (i >= i_hi) break;
        const uint32_t key = gs_rand_u32(seed ^ 0x5eedull, step, (uint32_t)ckey, (uint32_t)i);
        if ((key & kmask) == kpref) {
          const int slot = atomicAdd(&s_hash[3], 1);
          if (slot < 2048) { cand[2 * slot] = (int)key; cand[2 * slot + 1] = i; }
          bi |= 1u << bit;
        } else if ((key & kmask) < kpref) {
          bb |= 1u << bit;
        }
      }
      m_below[w] = bb;
      m_in[w] = bi;
    }
    for (int i = i_lo + 128; i < i_hi; ++i) {      // (slices beyond 128 indices: candidates only, classes recomputed below)
      const uint32_t key = gs_rand_u32(seed ^ 0x5eedull, step, (uint32_t)ckey, (uint32_t)i);
      if ((key & kmask) == kpref) {
        const int slot = atomicAdd(&s_hash[3], 1);
        if (slot < 2048) { cand[2 * slot] = (int)key; cand[2 * slot + 1] = i; }
      }
    }
    __syncthreads();
    const int m = min(s_hash[3], 2048);             // (> 2048 only for pools beyond 2^35 entries: the batch then comes out short: B below)
    for (int a = tid; a < m; a += 256) {            // rank of candidate a among the members of the crossing bin
      const uint32_t ka = (uint32_t)cand[2 * a];
      const int ia = cand[2 * a + 1];
      int rank = 0;
      for (int q = 0; q < m; ++q) {
        const uint32_t kq = (uint32_t)cand[2 * q];
        rank += (kq < ka) || (kq == ka && cand[2 * q + 1] < ia);
      }
      take[a] = rank < need;
    }
    __syncthreads();
    auto taken_in_bin = [&](int i) {
      bool tk = false;
      for (int q = 0; q < m; ++q) tk |= (cand[2 * q + 1] == i) && take[q];
      return tk;
    };
    int mine = 0, taken_total = 0;
    for (int pass = 0; pass < 2; ++pass) {          // count, then write in ascending pool order
      int base = 0;
      if (pass == 1) {
        base = block_excl_scan_256(mine, s_tmp, &taken_total);
        mine = 0;
      }
      if (use_mask) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          uint32_t in_w = m_in[w], tk_w = m_below[w];
          while (in_w) {                            // members of the crossing bin (few): look their decision up
            const int bit = __builtin_ctz(in_w);
            in_w &= in_w - 1u;
            if (taken_in_bin(i_lo + 32 * w + bit)) tk_w |= 1u << bit;
          }
          while (tk_w) {                            // ascending bit order = ascending pool order
            const int bit = __builtin_ctz(tk_w);
            tk_w &= tk_w - 1u;
            if (pass == 1 && base + mine < B) s_seed[base + mine] = i_lo + 32 * w + bit;
            ++mine;
          }
        }
      } else {
        for (int i = i_lo; i < i_hi; ++i) {
          const uint32_t key = gs_rand_u32(seed ^ 0x5eedull, step, (uint32_t)ckey, (uint32_t)i);
          const bool tk = ((key & kmask) == kpref) ? taken_in_bin(i) : (key & kmask) < kpref;   // top 12 or 24 key bits compared as a number
          if (tk) {
            if (pass == 1 && base + mine < B) s_seed[base + mine] = i;
            ++mine;
          }
        }
      }
    }
    __syncthreads();
    B = min(B, taken_total);                        // == B by construction (exact ranks); never read seed slots that were not written
    if (tid < B) s_seed[tid] = pool[s_seed[tid]];
  }
  __syncthreads();

  if (seeds_only) {      // full-neighbour mode (gs_gcond_set_full_neighbour): only the seed draw is used — same seeds as the sampled mode
    if (tid < B) blk.nid1[(long)c * cap1 + tid] = s_seed[tid];
    if (tid == 0) { blk.n_seed[c] = B; blk.n1[c] = B; }
    return;
  }
  // ---- hop 1: thread b samples the neighbours of seed b ------------------------------------------
  int k = 0, e0 = 0, v = 0;
  if (tid < B) {
    v = s_seed[tid];
    e0 = rowptr[v];
    int deg = rowptr[v + 1] - e0;
    k = sample_row(deg, f1, s_sel + tid * GS_MAX_F, seed ^ 0x1111ull, step, (uint32_t)(ckey * GS_MAX_BCAP + tid));
    mk_store(&mk[v], tid - GS_MARK_BIAS);          // seeds take local ids 0..B-1
    blk.nid1[(long)c * cap1 + tid] = v;
  }
  for (int j = 0; j < f1; ++j) {
    long slot = ((long)c * bcap + tid) * f1 + j;
    if (tid < B && j < k) {
      int e = e0 + s_sel[tid * GS_MAX_F + j];
      s_g[tid * GS_MAX_F + j] = col[e];
      blk.val1[slot] = val[e];
    } else if (tid < bcap) {
      if (tid < GS_MAX_BCAP) s_g[tid * GS_MAX_F + j] = -1;
      blk.val1[slot] = 0.f;
      blk.col1[slot] = 0;
    }
  }
  __syncthreads();
  // first occurrence (lowest edge-slot id) of every non-seed node
  for (int j = 0; j < k; ++j) atomicMin(&mk[s_g[tid * GS_MAX_F + j]], tid * f1 + j);
  __syncthreads();
  int nfirst = 0;
  for (int j = 0; j < k; ++j) nfirst += (mk_load(&mk[s_g[tid * GS_MAX_F + j]]) == tid * f1 + j);
  int total_new = 0;
  int base = block_excl_scan_256(nfirst, s_tmp, &total_new);
  {
    int run = 0;
    for (int j = 0; j < k; ++j) {
      int g = s_g[tid * GS_MAX_F + j];
      if (mk_load(&mk[g]) == tid * f1 + j) {
        int lid = B + base + run++;
        blk.nid1[(long)c * cap1 + lid] = g;
        mk_store(&mk[g], lid - GS_MARK_BIAS);
      }
    }
  }
  __syncthreads();
  const int n1 = B + total_new;
  for (int i = tid; i <= cap1; i += 256) s_cnt[i] = 0;
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    int lid = mk_load(&mk[s_g[tid * GS_MAX_F + j]]) + GS_MARK_BIAS;
    blk.col1[((long)c * bcap + tid) * f1 + j] = lid;
    s_g[tid * GS_MAX_F + j] = lid;   // the global id is no longer needed
    atomicAdd(&s_cnt[lid], 1);
  }
  __syncthreads();
  // ---- restore the marker (every touched entry back to FREE) --------------------------------------
  if (tid < B) mk_store(&mk[v], GS_MARK_FREE);
  for (int i = B + tid; i < n1; i += 256) mk_store(&mk[blk.nid1[(long)c * cap1 + i]], GS_MARK_FREE);

  // ---- CSC of the hop-1 block: exclusive scan of in-degrees, fill, sort each list ----------------
  const int per = (cap1 + 1 + 255) / 256;
  int lo = tid * per, hi = min(lo + per, cap1 + 1), lsum = 0;
  for (int i = lo; i < hi; ++i) lsum += s_cnt[i];
  int tot = 0;
  int off = block_excl_scan_256(lsum, s_tmp, &tot);
  int* tptr = blk.tptr1 + (long)c * (cap1 + 1);
  int* tedge = blk.tedge1 + (long)c * bcap * f1;
  for (int i = lo; i < hi; ++i) {
    int cnt = s_cnt[i];
    tptr[i] = off;
    s_cnt[i] = off;     // becomes the fill cursor
    off += cnt;
  }
  __syncthreads();
  // The lists are filled and sorted in LDS (s_sel: Floyd's scratch is no longer needed) and written out once, coalesced.  The first
  // form filled and insertion-sorted them in global memory: a dependent ~1 us round trip per move, 8 - 80 us per class depending on
  // its longest list (round-3 phase stamps), the longest part of the kernel for small pools whose seeds share neighbours.
  int* ledge = s_sel;
  for (int j = 0; j < k; ++j) {
    int lid = s_g[tid * GS_MAX_F + j];
    int pos = atomicAdd(&s_cnt[lid], 1);
    ledge[pos] = tid * f1 + j;
  }
  __syncthreads();
  for (int l = tid; l < n1; l += 256) {   // ascending edge-slot order -> deterministic summation order
    const int a = l ? s_cnt[l - 1] : 0, b = s_cnt[l];      // after the fill the cursor of list l stands at its end = the start of list l + 1
    for (int i = a + 1; i < b; ++i) {
      int x = ledge[i], q = i - 1;
      while (q >= a && ledge[q] > x) { ledge[q + 1] = ledge[q]; --q; }
      ledge[q + 1] = x;
    }
  }
  __syncthreads();
  for (int i = tid; i < tot; i += 256) tedge[i] = ledge[i];
  if (tid == 0) {
    blk.n_seed[c] = B;
    blk.n1[c] = n1;
  }
}

// the frontier the outermost hop samples from: hop 1's, or the last middle hop's (nlayers >= 3)
__device__ __forceinline__ void last_frontier(const gs_blocks& blk, const int*& n_last, const int*& nid_last, int& cap_last) {
  if (blk.nmid > 0) { n_last = blk.nm[blk.nmid - 1]; nid_last = blk.nidm[blk.nmid - 1]; cap_last = blk.capm[blk.nmid - 1]; }
  else { n_last = blk.n1; nid_last = blk.nid1; cap_last = blk.cap1; }
}

__global__ __launch_bounds__(256) void gs_sample_hop2_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                             const float* __restrict__ val, gs_blocks blk, uint64_t seed,
                                                             uint32_t step, int class_key0) {
  const int c = blockIdx.y, tid = threadIdx.x;
  const int l = blockIdx.x * 256 + tid;
  const int f2 = blk.f2;
  const int* n_last; const int* nid_last; int cap1;
  last_frontier(blk, n_last, nid_last, cap1);
  __shared__ int s_sel[256 * GS_MAX_F];
  if (l >= cap1) return;
  const long base = ((long)c * cap1 + l) * f2;
  if (l >= n_last[c]) {
    for (int j = 0; j < f2; ++j) { blk.col0[base + j] = 0; blk.val0[base + j] = 0.f; }
    return;
  }
  const int v = nid_last[(long)c * cap1 + l];
  const int e0 = rowptr[v], deg = rowptr[v + 1] - e0;
  // (two hops: the key of rounds 1-4; more hops: rows beyond 65 535 exist, the class moves into the seed)
  const int k = blk.nmid > 0 ? sample_row(deg, f2, s_sel + tid * GS_MAX_F, seed ^ 0x2222ull ^ ((uint64_t)(class_key0 + c) << 40), step, (uint32_t)l)
                             : sample_row(deg, f2, s_sel + tid * GS_MAX_F, seed ^ 0x2222ull, step, (uint32_t)((class_key0 + c) * 65536 + l));
  for (int j = 0; j < f2; ++j) {
    if (j < k) {
      int e = e0 + s_sel[tid * GS_MAX_F + j];
      blk.col0[base + j] = col[e];
      blk.val0[base + j] = val[e];
    } else {
      blk.col0[base + j] = 0;
      blk.val0[base + j] = 0.f;
    }
  }
}

// ---- a MIDDLE hop (nlayers >= 3; dataset/loader.py:205-210): frontier m - 1 (hop 1's for m = 0) -> frontier m ---------------------------
// One workgroup per class.  Every node of the previous frontier keeps <= fm neighbours (Floyd, like the other hops); the new frontier
// holds the previous one first (same local ids) and then every new node in the order of its FIRST edge slot (row-major over the
// block: the order PyG's relabelling produces) — a global atomicMin of slot indices on the class's marker row, then chunked exclusive
// scans over the slots; the CSC of the block (lists ascending by slot: a fixed summation order for the transposed product) is built
// in global memory with the per-level scratch `scrm`.  Frontiers of 4 096 .. 270 000 nodes do not fit LDS: everything is global.
__global__ __launch_bounds__(256) void gs_sample_mid_kernel(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                            const float* __restrict__ val, int n_nodes, int* marker, gs_blocks blk, int m,
                                                            uint64_t seed, uint32_t step, int class_key0) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int f = blk.fm[m], capc = blk.capm[m];
  const int capp = m == 0 ? blk.cap1 : blk.capm[m - 1];
  const int np = (m == 0 ? blk.n1 : blk.nm[m - 1])[c];
  const int* nidp = (m == 0 ? blk.nid1 : blk.nidm[m - 1]) + (long)c * capp;
  int* colm = blk.colm[m] + (long)c * capp * f;
  float* valm = blk.valm[m] + (long)c * capp * f;
  int* nidc = blk.nidm[m] + (long)c * capc;
  int* tptr = blk.tptrm[m] + (long)c * (capc + 1);
  int* tedge = blk.tedgem[m] + (long)c * capp * f;
  int* cur = blk.scrm[m] + (long)c * (capc + 1);
  int* mk = marker + (long)c * n_nodes;
  __shared__ int s_tmp[4];
  __shared__ int s_carry;
  const uint64_t sd = seed ^ (0x3333ull + (uint64_t)m) ^ ((uint64_t)(class_key0 + c) << 40);
  // 1. the targets keep their local ids; 2. every target row samples its neighbours (global ids for now)
  for (int r = tid; r < capp; r += 256) {
    if (r < np) {
      const int v = nidp[r];
      nidc[r] = v;
      mk_store(&mk[v], r - GS_MARK_BIAS);
      const int e0 = rowptr[v], deg = rowptr[v + 1] - e0;
      int sel[GS_MAX_F];
      const int k = sample_row(deg, f, sel, sd, step, (uint32_t)r);
      for (int j = 0; j < f; ++j) {
        if (j < k) { colm[(long)r * f + j] = col[e0 + sel[j]]; valm[(long)r * f + j] = val[e0 + sel[j]]; }
        else { colm[(long)r * f + j] = -1; valm[(long)r * f + j] = 0.f; }
      }
    } else {
      for (int j = 0; j < f; ++j) { colm[(long)r * f + j] = -1; valm[(long)r * f + j] = 0.f; }
    }
  }
  __threadfence();
  __syncthreads();
  // 3. first occurrence (lowest slot) of every node that is not a target
  const long nslots = (long)np * f;
  for (long sl = tid; sl < nslots; sl += 256) {
    const int g = colm[sl];
    if (g >= 0) atomicMin(&mk[g], (int)sl);
  }
  __threadfence();
  __syncthreads();
  // 4. new local ids in slot order (chunks of 256 slots, running carry)
  if (tid == 0) s_carry = np;
  __syncthreads();
  for (long base = 0; base < nslots; base += 256) {
    const long sl = base + tid;
    int g = -1, win = 0;
    if (sl < nslots) {
      g = colm[sl];
      win = g >= 0 && mk_load(&mk[g]) == (int)sl;
    }
    int tot = 0;
    const int ex = block_excl_scan_256(win, s_tmp, &tot);
    const int carry = s_carry;
    if (win) {
      const int lid = carry + ex;
      nidc[lid] = g;
      mk_store(&mk[g], lid - GS_MARK_BIAS);
    }
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  const int nc = s_carry;
  __threadfence();
  __syncthreads();
  // 5. local ids into the block; in-degree counts
  for (int i = tid; i <= capc; i += 256) { cur[i] = 0; }
  __threadfence();
  __syncthreads();
  for (long sl = tid; sl < (long)capp * f; sl += 256) {
    const int g = sl < nslots ? colm[sl] : -1;
    if (g >= 0 && valm[sl] != 0.f) {
      const int lid = mk_load(&mk[g]) + GS_MARK_BIAS;
      colm[sl] = lid;
      atomicAdd(&cur[lid], 1);
    } else {
      colm[sl] = 0;      // padding: (0, 0.f) like the other blocks
    }
  }
  __threadfence();
  __syncthreads();
  // 6. the marker row goes back to FREE
  for (int i = tid; i < nc; i += 256) mk_store(&mk[nidc[i]], GS_MARK_FREE);
  // 7. CSC: exclusive scan of the counts (chunks of 256 nodes), fill with atomic cursors, sort every list ascending
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base <= capc; base += 256) {
    const int i = base + tid;
    const int cnt = i <= capc ? cur[i] : 0;
    int tot = 0;
    const int ex = block_excl_scan_256(cnt, s_tmp, &tot);
    const int carry = s_carry;
    if (i <= capc) { tptr[i] = carry + ex; cur[i] = carry + ex; }
    __syncthreads();
    if (tid == 0) s_carry = carry + tot;
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  for (long sl = tid; sl < nslots; sl += 256) {
    // (a live slot is recognised by its weight: padding carries exactly 0.f, an edge of the normalised adjacency never does)
    if (valm[sl] != 0.f) {
      const int pos = atomicAdd(&cur[colm[sl]], 1);
      tedge[pos] = (int)sl;
    }
  }
  __threadfence();
  __syncthreads();
  for (int l = tid; l < nc; l += 256) {
    const int a = tptr[l], b = tptr[l + 1];
    for (int i = a + 1; i < b; ++i) {
      const int x = tedge[i];
      int q = i - 1;
      while (q >= a && tedge[q] > x) { tedge[q + 1] = tedge[q]; --q; }
      tedge[q + 1] = x;
    }
  }
  if (tid == 0) blk.nm[m][c] = nc;
}

// ---- distinct sources of the OUTERMOST block (round 6) ----------------------------------------------------------------------------
// The outermost hop keeps global source ids and is not de-duplicated (PyG's n_id is: dataset/loader.py:216-223) — the forward only
// gathers rows.  The backward's edge product gW1_c = sum_e x_src(e) (x) (w_e [H1[src(e)] > 0] Y[row e]) however depends on the SOURCE
// only through x and the mask: summing a source's in-edges first makes its K dimension the DISTINCT sources (ogbn-arxiv: 8 700 live
// slots, ~3 200 distinct per class).  This kernel derives, for the blocks of one class per workgroup: src0 = the distinct sources in
// the order of their first edge slot (deterministic), and the CSC of the block over them (sptr0 / sedge0: slots ascending per source —
// a fixed summation order).  Same marker protocol as the middle hops (row all FREE on entry and on exit); everything in global memory.
#define GS_DD_THREADS 1024
__device__ __forceinline__ int block_excl_scan_1024(int v, int* s_w, int* total) {      // s_w: 16 ints of LDS
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  __syncthreads();      // (s_w may still be read from the previous call)
  if (lane == 63) s_w[w] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < GS_DD_THREADS / 64; ++i) {
    const int t = s_w[i];
    if (i < w) base += t;
    tot += t;
  }
  *total = tot;
  return base + x - v;
}

// One workgroup of 1 024 threads per class; the slot / source loops are independent iterations (the loads of several are in flight),
// the two scans run over one contiguous chunk per thread.  (The first form — 256 threads, one scan step per 256 slots — took 423 us per
// sampling at ogbn-arxiv: profiles/r06/edge_src_materialised_not_kept.txt.)
__global__ __launch_bounds__(GS_DD_THREADS) void gs_dedup_last_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                                       const int* __restrict__ n_last, int cap_last, int f2, int n_nodes,
                                                                       int* marker, int* __restrict__ n0, int* __restrict__ src0,
                                                                       int* __restrict__ sptr0, int* __restrict__ sedge0,
                                                                       int* __restrict__ cur0, int cap0, int* __restrict__ nent,
                                                                       int* __restrict__ ent_src, int2* __restrict__ ent_row,
                                                                       float2* __restrict__ ent_w) {
  const int c = blockIdx.x, tid = threadIdx.x;
  const int* colc = col0 + (long)c * cap_last * f2;
  const float* valc = val0 + (long)c * cap_last * f2;
  int* mk = marker + (long)c * n_nodes;
  int* src = src0 + (long)c * cap0;
  int* sptr = sptr0 + (long)c * (cap0 + 1);
  int* sedge = sedge0 + (long)c * cap0;
  int* cur = cur0 + (long)c * (cap0 + 1);
  __shared__ int s_w[GS_DD_THREADS / 64];
  const int nslots = n_last[c] * f2;
  // 1. first occurrence (lowest slot) of every source (a live slot is recognised by its weight: padding carries exactly 0.f)
  for (int sl = tid; sl < nslots; sl += GS_DD_THREADS)
    if (valc[sl] != 0.f) atomicMin(&mk[colc[sl]], sl);
  __threadfence();
  __syncthreads();
  // 2. local ids in slot order: thread t owns the contiguous slots [t per, (t + 1) per)
  const int per = (nslots + GS_DD_THREADS - 1) / GS_DD_THREADS;
  const int lo = min(tid * per, nslots), hi = min(lo + per, nslots);
  unsigned winmask = 0;      // (per <= 32: cap_last f2 <= 32 768 slots; checked by the launcher)
  for (int sl = lo; sl < hi; ++sl) {
    const bool win = valc[sl] != 0.f && mk_load(&mk[colc[sl]]) == sl;
    winmask |= (win ? 1u : 0u) << (sl - lo);
  }
  int nc = 0;
  int lid = block_excl_scan_1024(__popc(winmask), s_w, &nc);
  for (int sl = lo; sl < hi; ++sl) {
    if ((winmask >> (sl - lo)) & 1u) {
      const int g = colc[sl];
      src[lid] = g;
      mk_store(&mk[g], lid - GS_MARK_BIAS);
      ++lid;
    }
  }
  for (int i = tid; i <= nc; i += GS_DD_THREADS) cur[i] = 0;
  __threadfence();
  __syncthreads();
  // 3. in-degree counts
  for (int sl = tid; sl < nslots; sl += GS_DD_THREADS)
    if (valc[sl] != 0.f) atomicAdd(&cur[mk_load(&mk[colc[sl]]) + GS_MARK_BIAS], 1);
  __threadfence();
  __syncthreads();
  // 4. exclusive scan of the counts (one contiguous chunk of sources per thread); cur becomes the fill cursor.  The same chunks carry
  //    the ENTRY scan: a source with k in-edges becomes ceil(k / 2) entries of the edge product (two edges each, below)
  int ebase = 0, el0 = 0, el1 = 0;
  {
    const int perl = (nc + 1 + GS_DD_THREADS - 1) / GS_DD_THREADS;
    const int l0 = min(tid * perl, nc + 1), l1 = min(l0 + perl, nc + 1);
    int sum = 0, esum = 0;
    for (int i = l0; i < l1; ++i) {
      const int cnt = i < nc ? cur[i] : 0;
      sum += cnt;
      esum += (cnt + 1) >> 1;
    }
    int tot = 0, etot = 0;
    int off = block_excl_scan_1024(sum, s_w, &tot);
    ebase = block_excl_scan_1024(esum, s_w, &etot);
    el0 = l0; el1 = min(l1, nc);
    for (int i = l0; i < l1; ++i) {
      const int cnt = i < nc ? cur[i] : 0;
      sptr[i] = off;
      cur[i] = off;
      off += cnt;
    }
    if (tid == 0 && nent) nent[c] = etot;
  }
  __threadfence();
  __syncthreads();
  // 5. fill, then the marker row goes back to FREE, then every list ascending
  for (int sl = tid; sl < nslots; sl += GS_DD_THREADS) {
    if (valc[sl] != 0.f) {
      const int pos = atomicAdd(&cur[mk_load(&mk[colc[sl]]) + GS_MARK_BIAS], 1);
      sedge[pos] = sl;
    }
  }
  __threadfence();
  __syncthreads();
  for (int i = tid; i < nc; i += GS_DD_THREADS) mk_store(&mk[src[i]], GS_MARK_FREE);
  for (int l = tid; l < nc; l += GS_DD_THREADS) {
    const int a = sptr[l], b = sptr[l + 1];
    for (int i = a + 1; i < b; ++i) {
      const int x = sedge[i];
      int q = i - 1;
      while (q >= a && sedge[q] > x) { sedge[q + 1] = sedge[q]; --q; }
      sedge[q + 1] = x;
    }
  }
  if (tid == 0) n0[c] = nc;
  // 6. the edge product's K entries (gs_edge.hip, ENT form): entry = (source, two of its in-edges as (row of Y, weight); the second
  //    weight is 0.f for an odd tail) in source order, a source's edges in ascending slot order — each thread emits its own chunk of
  //    sources (the lists it reads were sorted by OTHER threads: barrier first)
  if (nent) {
    __threadfence();
    __syncthreads();
    int* es = ent_src + (long)c * cap0;
    int2* er = ent_row + (long)c * cap0;
    float2* ew = ent_w + (long)c * cap0;
    int e = ebase;
    for (int l = el0; l < el1; ++l) {
      const int a = sptr[l], b = sptr[l + 1], g = src[l];
      for (int i = a; i < b; i += 2, ++e) {
        const int s0 = sedge[i], s1 = i + 1 < b ? sedge[i + 1] : s0;
        es[e] = g;
        er[e] = make_int2(s0 / f2, s1 / f2);
        ew[e] = make_float2(valc[s0], i + 1 < b ? valc[s1] : 0.f);
      }
    }
  }
}

// ---- the same structure by a chain of chip-wide launches (round 6) --------------------------------------------------------------------
// The one-workgroup-per-class kernel above takes 373 us per sampling at ogbn-arxiv (40 workgroups, serial scans over 14 080 slots) and its
// 1 024-thread workgroups hold compute units the persistent products of the main chain count on.  Here every phase is a launch over
// (chunks of 1 024 slots or sources) x classes, 256 threads each; the two scans are two-level (chunk sums, then chunk base + in-chunk
// scan).  Bitwise the structure of the kernel above: local ids in the order of the first edge slot, slots ascending per source, entries
// in source order.  scratch (ints): [C][GS_DDM_CH] first-slot counts per slot chunk | [C][GS_DDM_CH] in-edge sums per source chunk |
// [C][GS_DDM_CH] entry sums per source chunk | [C][cap0 + 1] entry offset per source | [C] number of LONG lists | [C][cap0 / GS_DDM_SMALL
// + 1] their sources.  A source sampled by many frontier rows (a hub) has a long slot list: lists of more than GS_DDM_SMALL slots are
// sorted by a workgroup in LDS (bitonic) instead of by one thread (the one-thread insertion sort of a 1 000-slot list was most of the
// first form's 373 us).
#define GS_DDM_CHUNK 1024
#define GS_DDM_CH 64      // chunks per class at most (65 536 slots)
#define GS_DDM_SMALL 16
#define GS_DDM_LDS 16384  // ints: the longest list the workgroup sort holds
__device__ __forceinline__ long ddm_off_eoff(int C) { return 3L * C * GS_DDM_CH; }
__device__ __forceinline__ long ddm_off_nbig(int C, int cap0) { return 3L * C * GS_DDM_CH + (long)C * (cap0 + 1); }
__device__ __forceinline__ long ddm_off_big(int C, int cap0) { return ddm_off_nbig(C, cap0) + C; }
__device__ __forceinline__ int ddm_big_cap(int cap0) { return cap0 / GS_DDM_SMALL + 1; }
// 1. first occurrence (lowest slot) of every source; cur <- 0
__global__ __launch_bounds__(256) void gs_ddm_first_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                           const int* __restrict__ n_last, int cap_last, int f2, int n_nodes, int* marker,
                                                           int* __restrict__ cur0, int cap0, int* __restrict__ scratch) {
  const int c = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
  if (sl <= cap0) cur0[(long)c * (cap0 + 1) + sl] = 0;
  if (sl == 0) scratch[ddm_off_nbig(gridDim.y, cap0) + c] = 0;
  if (sl >= n_last[c] * f2) return;
  const long o = (long)c * cap_last * f2 + sl;
  if (val0[o] != 0.f) atomicMin(&marker[(long)c * n_nodes + col0[o]], sl);
}
// 2. / 3. slots that are their source's first: count per chunk (assign = 0), then local ids in slot order (assign = 1)
template <int ASSIGN>
__global__ __launch_bounds__(256) void gs_ddm_ids_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                         const int* __restrict__ n_last, int cap_last, int f2, int n_nodes, int* marker,
                                                         int* __restrict__ n0, int* __restrict__ src0, int cap0, int* __restrict__ scratch) {
  __shared__ int s_w[4];
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int nslots = n_last[c] * f2;
  const int nch = (nslots + GS_DDM_CHUNK - 1) / GS_DDM_CHUNK;
  if (ch >= nch && !(ASSIGN && ch == 0)) return;
  const int* colc = col0 + (long)c * cap_last * f2;
  const float* valc = val0 + (long)c * cap_last * f2;
  int* mk = marker + (long)c * n_nodes;
  int* cnt = scratch + (long)c * GS_DDM_CH;
  const int lo = ch * GS_DDM_CHUNK + tid * 4;
  unsigned win = 0;
  int g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int sl = lo + j;
    g[j] = 0;
    if (sl < nslots && valc[sl] != 0.f) {
      g[j] = colc[sl];
      win |= (mk_load(&mk[g[j]]) == sl ? 1u : 0u) << j;
    }
  }
  int tot = 0;
  int lid = block_excl_scan_256(__popc(win), s_w, &tot);
  if (!ASSIGN) {
    if (tid == 0) cnt[ch] = tot;
    return;
  }
  int base = 0, all = 0;
  for (int i = 0; i < nch; ++i) {
    const int t = cnt[i];
    if (i < ch) base += t;
    all += t;
  }
  if (ch == 0 && tid == 0) n0[c] = all;
  lid += base;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if ((win >> j) & 1u) {
      src0[(long)c * cap0 + lid] = g[j];
      mk_store(&mk[g[j]], lid - GS_MARK_BIAS);      // (every other slot of this source compares its own index with the marker: no longer equal)
      ++lid;
    }
}
// 4. / 6. in-degree counts (FILL = 0) and the slot lists (FILL = 1: cur is the fill cursor)
template <int FILL>
__global__ __launch_bounds__(256) void gs_ddm_edges_kernel(const int* __restrict__ col0, const float* __restrict__ val0,
                                                           const int* __restrict__ n_last, int cap_last, int f2, int n_nodes,
                                                           const int* marker, int* __restrict__ cur0, int* __restrict__ sedge0, int cap0) {
  const int c = blockIdx.y, sl = blockIdx.x * 256 + threadIdx.x;
  if (sl >= n_last[c] * f2) return;
  const long o = (long)c * cap_last * f2 + sl;
  if (val0[o] == 0.f) return;
  const int l = mk_load(&marker[(long)c * n_nodes + col0[o]]) + GS_MARK_BIAS;
  const int pos = atomicAdd(&cur0[(long)c * (cap0 + 1) + l], 1);
  if (FILL) sedge0[(long)c * cap0 + pos] = sl;
}
// 5. exclusive scans over the sources: in-edge counts -> sptr / fill cursor, ceil(count / 2) -> entry offsets.  WRITE = 0: chunk sums
template <int WRITE>
__global__ __launch_bounds__(256) void gs_ddm_scan_kernel(const int* __restrict__ n0, int* __restrict__ sptr0, int* __restrict__ cur0,
                                                          int cap0, int* __restrict__ nent, int* __restrict__ scratch, int C) {
  __shared__ int s_w[4];
  const int c = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int nc = n0[c];
  const int nch = (nc + 1 + GS_DDM_CHUNK - 1) / GS_DDM_CHUNK;      // nc + 1 entries: sptr[nc] closes the last list
  if (ch >= nch) return;
  int* cur = cur0 + (long)c * (cap0 + 1);
  int* sums = scratch + (long)C * GS_DDM_CH + (long)c * GS_DDM_CH;
  int* esums = scratch + 2L * C * GS_DDM_CH + (long)c * GS_DDM_CH;
  int* eoff = scratch + 3L * C * GS_DDM_CH + (long)c * (cap0 + 1);
  const int lo = ch * GS_DDM_CHUNK + tid * 4;
  int cnt[4], sum = 0, esum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    cnt[j] = lo + j < nc ? cur[lo + j] : 0;
    sum += cnt[j];
    esum += (cnt[j] + 1) >> 1;
  }
  int tot = 0, etot = 0;
  int off = block_excl_scan_256(sum, s_w, &tot);
  int eo = block_excl_scan_256(esum, s_w, &etot);
  if (!WRITE) {
    if (tid == 0) { sums[ch] = tot; esums[ch] = etot; }
    return;
  }
  int base = 0, ebase = 0, eall = 0;
  for (int i = 0; i < nch; ++i) {
    if (i < ch) { base += sums[i]; ebase += esums[i]; }
    eall += esums[i];
  }
  if (ch == 0 && tid == 0 && nent) nent[c] = eall;
  off += base; eo += ebase;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lo + j <= nc) {
      sptr0[(long)c * (cap0 + 1) + lo + j] = off;
      cur[lo + j] = off;
      eoff[lo + j] = eo;
      off += cnt[j];
      eo += (cnt[j] + 1) >> 1;
      if (cnt[j] > GS_DDM_SMALL) {      // (the order of this list does not matter: each of its sources is finished on its own)
        const int pos = atomicAdd(&scratch[ddm_off_nbig(C, cap0) + c], 1);
        scratch[ddm_off_big(C, cap0) + (long)c * ddm_big_cap(cap0) + pos] = lo + j;
      }
    }
}
// 7. per source: the marker goes back to FREE, its slot list ascending, its entries
__global__ __launch_bounds__(256) void gs_ddm_finish_kernel(const float* __restrict__ val0, int cap_last, int f2, int n_nodes, int* marker,
                                                            const int* __restrict__ n0, const int* __restrict__ src0,
                                                            const int* __restrict__ sptr0, int* __restrict__ sedge0, int cap0,
                                                            const int* __restrict__ scratch, int C, int* __restrict__ ent_src,
                                                            int2* __restrict__ ent_row, float2* __restrict__ ent_w) {
  const int c = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
  if (l >= n0[c]) return;
  const int g = src0[(long)c * cap0 + l];
  mk_store(&marker[(long)c * n_nodes + g], GS_MARK_FREE);
  int* sedge = sedge0 + (long)c * cap0;
  const int a = sptr0[(long)c * (cap0 + 1) + l], b = sptr0[(long)c * (cap0 + 1) + l + 1];
  if (b - a > GS_DDM_SMALL) return;      // a long list: gs_ddm_long_kernel
  for (int i = a + 1; i < b; ++i) {
    const int x = sedge[i];
    int q = i - 1;
    while (q >= a && sedge[q] > x) { sedge[q + 1] = sedge[q]; --q; }
    sedge[q + 1] = x;
  }
  if (!ent_src) return;
  const float* valc = val0 + (long)c * cap_last * f2;
  int e = scratch[ddm_off_eoff(C) + (long)c * (cap0 + 1) + l];
  for (int i = a; i < b; i += 2, ++e) {
    const int s0 = sedge[i], s1 = i + 1 < b ? sedge[i + 1] : s0;
    ent_src[(long)c * cap0 + e] = g;
    ent_row[(long)c * cap0 + e] = make_int2(s0 / f2, s1 / f2);
    ent_w[(long)c * cap0 + e] = make_float2(valc[s0], i + 1 < b ? valc[s1] : 0.f);
  }
}

// 8. the long lists: one workgroup per list at a time — bitonic sort in LDS, then its entries by all threads
__global__ __launch_bounds__(256) void gs_ddm_long_kernel(const float* __restrict__ val0, int cap_last, int f2, const int* __restrict__ src0,
                                                          const int* __restrict__ sptr0, int* __restrict__ sedge0, int cap0,
                                                          const int* __restrict__ scratch, int C, int* __restrict__ ent_src,
                                                          int2* __restrict__ ent_row, float2* __restrict__ ent_w) {
  __shared__ int s_v[GS_DDM_LDS];
  const int c = blockIdx.y, tid = threadIdx.x;
  const int nbig = scratch[ddm_off_nbig(C, cap0) + c];
  const int* big = scratch + ddm_off_big(C, cap0) + (long)c * ddm_big_cap(cap0);
  const float* valc = val0 + (long)c * cap_last * f2;
  int* sedge = sedge0 + (long)c * cap0;
  for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const int l = big[bi];
    const int a = sptr0[(long)c * (cap0 + 1) + l], k = sptr0[(long)c * (cap0 + 1) + l + 1] - a;
    const int* sorted = sedge + a;
    __syncthreads();      // (s_v of the previous list)
    if (k <= GS_DDM_LDS) {
      int n2 = 1;
      while (n2 < k) n2 <<= 1;
      for (int i = tid; i < n2; i += 256) s_v[i] = i < k ? sedge[a + i] : 0x7fffffff;
      __syncthreads();
      for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int t = tid; t < (n2 >> 1); t += 256) {
            const int i = 2 * t - (t & (stride - 1)), j = i + stride;      // i has bit `stride` clear
            const bool up = (i & size) == 0;
            const int x = s_v[i], y = s_v[j];
            if ((x > y) == up) { s_v[i] = y; s_v[j] = x; }
          }
          __syncthreads();
        }
      for (int i = tid; i < k; i += 256) sedge[a + i] = s_v[i];
      sorted = s_v;
    } else {      // (more than 16 384 of a class's slots on ONE source: one thread, heap sort in place — correct, slow, not expected)
      if (tid == 0) {
        int* v = sedge + a;
        for (int st = k / 2 - 1; st >= 0; --st)
          for (int r = st;;) { int ch = 2 * r + 1; if (ch >= k) break; if (ch + 1 < k && v[ch + 1] > v[ch]) ++ch; if (v[r] >= v[ch]) break; const int t = v[r]; v[r] = v[ch]; v[ch] = t; r = ch; }
        for (int e = k - 1; e > 0; --e) {
          const int t0 = v[0]; v[0] = v[e]; v[e] = t0;
          for (int r = 0;;) { int ch = 2 * r + 1; if (ch >= e) break; if (ch + 1 < e && v[ch + 1] > v[ch]) ++ch; if (v[r] >= v[ch]) break; const int t = v[r]; v[r] = v[ch]; v[ch] = t; r = ch; }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    if (ent_src) {
      const int g = src0[(long)c * cap0 + l];
      const int e0 = scratch[ddm_off_eoff(C) + (long)c * (cap0 + 1) + l];
      for (int j = tid; 2 * j < k; j += 256) {
        const int s0 = sorted[2 * j], s1 = 2 * j + 1 < k ? sorted[2 * j + 1] : s0;
        ent_src[(long)c * cap0 + e0 + j] = g;
        ent_row[(long)c * cap0 + e0 + j] = make_int2(s0 / f2, s1 / f2);
        ent_w[(long)c * cap0 + e0 + j] = make_float2(valc[s0], 2 * j + 1 < k ? valc[s1] : 0.f);
      }
    }
  }
}

static int g_dedup_form = 1;      // 1: the chain of chip-wide launches (needs scratch); 0: one workgroup per class (measurement / test hook)
extern "C" int gs_dedup_set_form(int form) {
  if (form < 0 || form > 1) GS_FAIL(GS_ERR_ARG, "gs_dedup_set_form: 0 or 1");
  g_dedup_form = form;
  return GS_OK;
}
extern "C" long gs_dedup_scratch_ints(int nclass, int cap0) {
  return 3L * nclass * GS_DDM_CH + (long)nclass * (cap0 + 1) + nclass + (long)nclass * (cap0 / GS_DDM_SMALL + 1);
}

// engine-internal (gs_engine.hip): the distinct-source structure of `nclass` classes' outermost blocks
int gs_dedup_last(const int* col0, const float* val0, const int* n_last, int cap_last, int f2, int n_nodes, int* marker, int nclass,
                  int* n0, int* src0, int* sptr0, int* sedge0, int* cur0, int cap0, int* nent, int* ent_src, void* ent_row, void* ent_w,
                  int* scratch, hipStream_t s) {
  if (nclass < 1) return GS_OK;
  if ((long)cap_last * f2 > 32L * GS_DD_THREADS) GS_FAIL(GS_ERR_UNSUPPORTED, "gs_dedup_last: more than %d edge slots per class", 32 * GS_DD_THREADS);
  if (g_dedup_form == 1 && scratch && cap0 == cap_last * f2 && cap0 <= GS_DDM_CH * GS_DDM_CHUNK) {
    const dim3 gslot(gs_cdiv(cap0 + 1, 256), nclass), gch(gs_cdiv(cap0 + 1, GS_DDM_CHUNK), nclass), gsrc(gs_cdiv(cap0, 256), nclass);
    hipLaunchKernelGGL(gs_ddm_first_kernel, gslot, dim3(256), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, cur0, cap0, scratch);
    hipLaunchKernelGGL((gs_ddm_ids_kernel<0>), gch, dim3(256), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, n0, src0, cap0, scratch);
    hipLaunchKernelGGL((gs_ddm_ids_kernel<1>), gch, dim3(256), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, n0, src0, cap0, scratch);
    hipLaunchKernelGGL((gs_ddm_edges_kernel<0>), gslot, dim3(256), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, cur0, sedge0, cap0);
    hipLaunchKernelGGL((gs_ddm_scan_kernel<0>), gch, dim3(256), 0, s, n0, sptr0, cur0, cap0, nent, scratch, nclass);
    hipLaunchKernelGGL((gs_ddm_scan_kernel<1>), gch, dim3(256), 0, s, n0, sptr0, cur0, cap0, nent, scratch, nclass);
    hipLaunchKernelGGL((gs_ddm_edges_kernel<1>), gslot, dim3(256), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, cur0, sedge0, cap0);
    hipLaunchKernelGGL(gs_ddm_finish_kernel, gsrc, dim3(256), 0, s, val0, cap_last, f2, n_nodes, marker, n0, src0, sptr0, sedge0, cap0, scratch,
                       nclass, nent ? ent_src : (int*)nullptr, (int2*)ent_row, (float2*)ent_w);
    hipLaunchKernelGGL(gs_ddm_long_kernel, dim3(16, nclass), dim3(256), 0, s, val0, cap_last, f2, src0, sptr0, sedge0, cap0, scratch, nclass,
                       nent ? ent_src : (int*)nullptr, (int2*)ent_row, (float2*)ent_w);
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  hipLaunchKernelGGL(gs_dedup_last_kernel, dim3(nclass), dim3(GS_DD_THREADS), 0, s, col0, val0, n_last, cap_last, f2, n_nodes, marker, n0, src0,
                     sptr0, sedge0, cur0, cap0, nent, ent_src, (int2*)ent_row, (float2*)ent_w);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

// members of the crossing key bin that are ranked directly (<= 2048: the LDS candidate list); larger bins get a second 12-bit
// level.  Test hook: a small value forces the two-level path on ordinary pools (same seeds either way: the selection is exact).
static int g_cand_cap = 2048;
extern "C" int gs_sample_set_cand_cap(int cap) {
  if (cap < 1 || cap > 2048) GS_FAIL(GS_ERR_ARG, "gs_sample_set_cand_cap: cap must be in [1, 2048]");
  g_cand_cap = cap;
  return GS_OK;
}

int gs_sample_blocks_ex(const int* rowptr, const int* col, const float* val, int n_nodes, const int* cls_ptr,
                        const int* cls_nodes, int* marker, const gs_blocks* out, uint64_t seed, uint32_t step, int class_key0,
                        gs_stream_t stream, int seeds_only) {
  if (!rowptr || !col || !val || !cls_ptr || !cls_nodes || !marker || !out) GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: null pointer");
  const gs_blocks& b = *out;
  if (b.bcap < 1 || b.bcap > GS_MAX_BCAP) GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: bcap must be in [1, %d]", GS_MAX_BCAP);
  if (b.f1 < 1 || b.f1 > GS_MAX_F || b.f2 < 1 || b.f2 > GS_MAX_F)
    GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: fan-outs must be in [1, %d]", GS_MAX_F);
  if (b.cap1 != b.bcap * (1 + b.f1)) GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: cap1 must equal bcap*(1+f1)");
  if (b.C < 1) return GS_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs_sample_hop1_kernel, dim3(b.C), dim3(256), 0, s, rowptr, col, val, n_nodes, cls_ptr, cls_nodes,
                     marker, b, seed, step, class_key0, seeds_only, g_cand_cap);
  if (seeds_only) {
    GS_LAUNCH_CHECK();
    return GS_OK;
  }
  if (b.nmid < 0 || b.nmid > GS_MAX_MID) GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: 0 <= nmid <= %d", GS_MAX_MID);
  int cap_last = b.cap1;
  for (int m = 0; m < b.nmid; ++m) {
    if (b.fm[m] < 1 || b.fm[m] > GS_MAX_F || b.capm[m] != cap_last * (1 + b.fm[m]))
      GS_FAIL(GS_ERR_ARG, "gs_sample_blocks: middle hop %d: fan-out in [1, %d], capm = cap(m-1) * (1 + fm)", m, GS_MAX_F);
    hipLaunchKernelGGL(gs_sample_mid_kernel, dim3(b.C), dim3(256), 0, s, rowptr, col, val, n_nodes, marker, b, m, seed, step, class_key0);
    cap_last = b.capm[m];
  }
  hipLaunchKernelGGL(gs_sample_hop2_kernel, dim3(gs_cdiv(cap_last, 256), b.C), dim3(256), 0, s, rowptr, col, val, b, seed,
                     step, class_key0);
  GS_LAUNCH_CHECK();
  return GS_OK;
}

extern "C" int gs_sample_blocks(const int* rowptr, const int* col, const float* val, int n_nodes, const int* cls_ptr,
                                const int* cls_nodes, int* marker, const gs_blocks* out, uint64_t seed, uint32_t step,
                                gs_stream_t stream) {
  return gs_sample_blocks_ex(rowptr, col, val, n_nodes, cls_ptr, cls_nodes, marker, out, seed, step, 0, stream, 0);
}
