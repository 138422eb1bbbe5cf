// pt_bvh_build.hip — SURVEY.md §8 f-4: the reference's binned-SAH BVH build (computeInfoForBVH +
// buildBVH + SAH + partitionPrimitives, main.cu:20-233; call site main.cu:524-530) on the GPU.
//
// "Reference-tree" mode: the output (pre-order BVHnode array + BVHindices permutation, largest
// leaf, backup count) is BYTE-IDENTICAL to the host builder's (novum_host.cpp: Builder; tests/
// test_bvh_build.py holds both to the CPU restatement). The reference recursion is depth-first and
// sequential; here every tree LEVEL is one batch of data-parallel passes over the primitive
// positions (a node's primitives are a contiguous range of the index array, as in the reference):
//
//   classify   per node      bounds -> node record; leaf test (main.cu:153); split axis (:161-168)
//   bin        per position  12 buckets over the node bounds on the axis (:76-85); unions and counts
//                            through order-independent atomics on order-preserving integer keys,
//                            staged in LDS per workgroup while a node is large
//   sah        per node      the reference's cost loop incl. its right-count quirk (:90-117)
//   (sort)     rare          median fallback (:119-128) = total-order sort of the node's range:
//                            rank by counting for small nodes, bitonic network for large ones
//   flag+scan  per position  centroid < splitPos (:185-188) and an exclusive prefix sum over ALL
//                            positions; numLeft of a node = S[end] - S[start]
//   (mean)     rare          centroid-mean retry (:192-202): a float sum in the reference's order,
//                            one wave per node, lanes fetch, the adds stay sequential
//   partition  per position  partitionPrimitives (:49-62) is a Lomuto loop; its result has a closed
//                            form: the k-th element below splitPos lands on start+k, and an element
//                            not below splitPos standing on slot p is moved, whenever p - start <
//                            numLeft, to the original position of the (p-start)-th element below
//                            splitPos — followed until p - start >= numLeft. The same pass unions
//                            the primitive's box into its child's bounds (min/max are exact, so the
//                            union order does not matter).
//
// Nodes are numbered breadth-first while building; two sweeps over the levels (subtree sizes up,
// pre-order index down) give the reference's depth-first numbering. One host synchronisation per
// level (after `sah`: it tells the host the level's node count and whether a sort is needed).
// All float arithmetic is written as the reference writes it and compiled with -ffp-contract=off
// (DESIGN.md §4). min/max over boxes use the total order of the integer keys, which is fminf/fmaxf
// on the finite values the builder accepts and puts -0 below +0 (the host builder does the same).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_device.h"

extern "C" int pt_fail_(int code, const char* msg);

namespace {

constexpr int NB = 12;
constexpr int kBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;
constexpr int kStageMin = 4096;       // nodes at least this large accumulate in LDS first (atomic contention)
constexpr int kChunks = 1024;         // workgroups of the per-position accumulation passes
constexpr int kSortSmall = 1024;      // median fallback: rank-by-counting up to here, bitonic network above
constexpr int kSortTile = 2048;       // keys one workgroup sorts in LDS
#ifndef PT_BVH_SMALL
#define PT_BVH_SMALL 64
#endif
constexpr int kSmallNode = PT_BVH_SMALL;   // a node of at most this many primitives (<= 64) is finished, whole subtree, by ONE WAVE; 0 = off

enum : int { ST_LEAF = 1, ST_SPLITTING = 2, ST_SORT = 3, ST_OK = 4, ST_REDO = 5, ST_SMALL = 6 };

struct WorkNode {                 // 64 B, one per node of the level being built
    int start, end, out, axis;
    unsigned lo[3], hi[3];        // bounds as order-preserving keys (atomicMin / atomicMax targets)
    float split;
    int numLeft, state, child, bin;
    int pad;
};
struct Bins { unsigned lo[NB][3], hi[NB][3]; int cnt[NB]; };
struct OutNode { float lo[3], hi[3]; int left, right, first, count, size, pre, sub, height, subInternal, pad; };   // sub >= 0: root of a subtree in Arrays::sub
struct SubNode { float lo[3], hi[3]; int left, right, first, count; };                                          // children as indices local to the subtree (pre-order)
struct SortRec { int w, start, end, axis; };
struct Ctl { int curCount, nextCount, outCount, binCount, anyRedo, redoCount, sortCount, sortMax, largestLeaf, backups, sortFallbacks, bad, subCount; };

__host__ __device__ inline unsigned fkey(float f) {
    unsigned u;
    __builtin_memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float funkey(unsigned k) {
    unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
__device__ inline unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ inline unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ inline unsigned wave_min(unsigned v) {
    for (int o = 32; o; o >>= 1) v = umin_(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
__device__ inline unsigned wave_max(unsigned v) {
    for (int o = 32; o; o >>= 1) v = umax_(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }
__device__ inline float comp3(const float* cx, const float* cy, const float* cz, int axis, int id) {
    return axis == 0 ? cx[id] : (axis == 1 ? cy[id] : cz[id]);
}

struct Arrays {
    const pt_float4* pos; int nPos;
    const pt_triangle* mesh; int n;
    float *cx, *cy, *cz;
    float4 *lo, *hi;
    int *idxA, *idxB, *nodeA, *nodeB, *tmp, *S, *g, *blockSum, *redoList;
    unsigned char* F;
    WorkNode *workA, *workB;
    Bins* bins;
    OutNode* out;
    pt_bvh_node* fin;
    SortRec* sortRec;
    SubNode* sub;
    unsigned long long* keys;
    Ctl* ctl;
};

// ---- accumulation of boxes into per-node slots ---------------------------------------------------
// A "slot" is one box (+ count) of a node: its 12 buckets (bin), its two children (scatter), or the
// root itself (prims). Small nodes take global atomics directly. A node of kStageMin primitives or
// more would put thousands of atomics on the same few words, so a workgroup walking its contiguous
// chunk of positions accumulates the node it is currently in in LDS and flushes once per node.
template <int S> struct StageMem { unsigned lo[S * 3], hi[S * 3]; int cnt[S]; int cand; };

template <int S> __device__ inline void stage_reset(StageMem<S>& m) {
    for (int t = threadIdx.x; t < S * 3; t += blockDim.x) { m.lo[t] = 0xffffffffu; m.hi[t] = 0u; }
    for (int t = threadIdx.x; t < S; t += blockDim.x) m.cnt[t] = 0;
}
template <int S, class Dest> __device__ inline void stage_flush(StageMem<S>& m, int cur, const Dest& dest) {
    __syncthreads();
    if (cur >= 0) {
        for (int t = threadIdx.x; t < S * 3; t += blockDim.x) {
            int slot = t / 3, k = t - 3 * slot;
            unsigned l = m.lo[t], h = m.hi[t];
            if (l != 0xffffffffu) atomicMin(dest.lo(cur, slot) + k, l);
            if (h != 0u) atomicMax(dest.hi(cur, slot) + k, h);
            m.lo[t] = 0xffffffffu; m.hi[t] = 0u;
        }
        for (int t = threadIdx.x; t < S; t += blockDim.x) {
            int c = m.cnt[t];
            if (c) { int* p = dest.cnt(cur, t); if (p) atomicAdd(p, c); }
            m.cnt[t] = 0;
        }
    }
    __syncthreads();
}
// One tile of the workgroup's chunk: lanes with `pend` add (kl, kh) to slot `slot` of node `node`.
// Must be called by every thread of the workgroup. `cur` is the node currently held in LDS.
template <int S, class Dest>
__device__ inline void stage_add(StageMem<S>& m, int& cur, bool pend, int node, int slot, const unsigned kl[3], const unsigned kh[3], const Dest& dest) {
    for (;;) {
        if (threadIdx.x == 0) m.cand = INT_MAX;
        __syncthreads();
        if (pend) atomicMin(&m.cand, node);
        __syncthreads();
        const int cand = m.cand;
        if (cand == INT_MAX) break;
        if (cand != cur) { stage_flush(m, cur, dest); cur = cand; }
        const bool sel = pend && node == cand;
        unsigned long long rem = __ballot(sel);
        while (rem) {
            int l = __ffsll((unsigned long long)rem) - 1;
            int s0 = __shfl(slot, l);
            bool ss = sel && slot == s0;
            unsigned long long sm = __ballot(ss);
            unsigned r[6];
            for (int k = 0; k < 3; k++) { r[k] = wave_min(ss ? kl[k] : 0xffffffffu); r[3 + k] = wave_max(ss ? kh[k] : 0u); }
            if (lane_id() == l) {
                for (int k = 0; k < 3; k++) { atomicMin(&m.lo[s0 * 3 + k], r[k]); atomicMax(&m.hi[s0 * 3 + k], r[3 + k]); }
                atomicAdd(&m.cnt[s0], (int)__popcll(sm));
            }
            rem &= ~sm;
        }
        pend = pend && !sel;
        __syncthreads();
    }
}
// Direct path for small nodes: when the whole wave works on one node, reduce per slot first.
template <class Dest>
__device__ inline void direct_add(bool act, int node, int slot, const unsigned kl[3], const unsigned kh[3], const Dest& dest) {
    unsigned long long am = __ballot(act);
    if (am == 0) return;
    int lead = __ffsll((unsigned long long)am) - 1;
    bool uniform = __all(!act || node == __shfl(node, lead)) != 0;
    if (uniform) {
        int n0 = __shfl(node, lead);
        unsigned long long rem = am;
        while (rem) {
            int l = __ffsll((unsigned long long)rem) - 1;
            int s0 = __shfl(slot, l);
            bool ss = act && slot == s0;
            unsigned long long sm = __ballot(ss);
            unsigned r[6];
            for (int k = 0; k < 3; k++) { r[k] = wave_min(ss ? kl[k] : 0xffffffffu); r[3 + k] = wave_max(ss ? kh[k] : 0u); }
            if (lane_id() == l) {
                for (int k = 0; k < 3; k++) { atomicMin(dest.lo(n0, s0) + k, r[k]); atomicMax(dest.hi(n0, s0) + k, r[3 + k]); }
                int* c = dest.cnt(n0, s0);
                if (c) atomicAdd(c, (int)__popcll(sm));
            }
            rem &= ~sm;
        }
    } else if (act) {
        for (int k = 0; k < 3; k++) { atomicMin(dest.lo(node, slot) + k, kl[k]); atomicMax(dest.hi(node, slot) + k, kh[k]); }
        int* c = dest.cnt(node, slot);
        if (c) atomicAdd(c, 1);
    }
}

struct RootDest {
    WorkNode* root;
    __device__ unsigned* lo(int, int) const { return root->lo; }
    __device__ unsigned* hi(int, int) const { return root->hi; }
    __device__ int* cnt(int, int) const { return nullptr; }
};
struct BinDest {
    const WorkNode* work; Bins* bins;
    __device__ unsigned* lo(int w, int b) const { return bins[work[w].bin].lo[b]; }
    __device__ unsigned* hi(int w, int b) const { return bins[work[w].bin].hi[b]; }
    __device__ int* cnt(int w, int b) const { return &bins[work[w].bin].cnt[b]; }
};
struct ChildDest {
    const WorkNode* work; WorkNode* next;
    __device__ unsigned* lo(int w, int c) const { return next[work[w].child + c].lo; }
    __device__ unsigned* hi(int w, int c) const { return next[work[w].child + c].hi; }
    __device__ int* cnt(int, int) const { return nullptr; }
};

// computeInfoForBVH, main.cu:20-47, plus the root's bounds and the identity permutation (:503-504).
__global__ __launch_bounds__(kBlock) void k_prims(Arrays A, int chunk) {
    __shared__ StageMem<1> sm;
    stage_reset(sm);
    __syncthreads();
    int cur = -1;
    const RootDest dest{A.workA};
    const bool staged = A.n >= kStageMin;
    const int b0 = blockIdx.x * chunk, b1 = min(A.n, b0 + chunk);
    for (int base = b0; base < b1; base += kBlock) {
        int i = base + threadIdx.x;
        bool ok = i < b1;
        unsigned kl[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kh[3] = {0u, 0u, 0u};
        if (ok) {
            pt_triangle t = A.mesh[i];
            int ia = t.aInd, ib = t.bInd, ic = t.cInd;
            if ((unsigned)ia >= (unsigned)A.nPos || (unsigned)ib >= (unsigned)A.nPos || (unsigned)ic >= (unsigned)A.nPos) { A.ctl->bad = 1; ia = ib = ic = 0; }
            pt_float4 a = A.pos[ia], b = A.pos[ib], c = A.pos[ic];
            float mx = fminf(fminf(a.x, b.x), c.x) - 0.000001f, my = fminf(fminf(a.y, b.y), c.y) - 0.000001f, mz = fminf(fminf(a.z, b.z), c.z) - 0.000001f;
            float Mx = fmaxf(fmaxf(a.x, b.x), c.x) + 0.000001f, My = fmaxf(fmaxf(a.y, b.y), c.y) + 0.000001f, Mz = fmaxf(fmaxf(a.z, b.z), c.z) + 0.000001f;
            A.cx[i] = (a.x + b.x + c.x) / 3.0f; A.cy[i] = (a.y + b.y + c.y) / 3.0f; A.cz[i] = (a.z + b.z + c.z) / 3.0f;
            A.lo[i] = make_float4(mx, my, mz, 0.0f); A.hi[i] = make_float4(Mx, My, Mz, 0.0f);
            // non-finite geometry has no defined tree in the reference either (NaN compares); refuse it
            float big = fmaxf(fmaxf(fabsf(mx), fabsf(my)), fmaxf(fmaxf(fabsf(mz), fabsf(Mx)), fmaxf(fabsf(My), fabsf(Mz))));
            float any = ((a.x + a.y + a.z) + (b.x + b.y + b.z)) + (c.x + c.y + c.z);          // NaN anywhere -> NaN
            if (!(big <= 1e37f) || any != any) A.ctl->bad = 2;
            A.idxA[i] = i; A.nodeA[i] = 0; A.nodeB[i] = -1;
            kl[0] = fkey(mx); kl[1] = fkey(my); kl[2] = fkey(mz); kh[0] = fkey(Mx); kh[1] = fkey(My); kh[2] = fkey(Mz);
        }
        if (staged) stage_add(sm, cur, ok, 0, 0, kl, kh, dest);
        else direct_add(ok, 0, 0, kl, kh, dest);
    }
    stage_flush(sm, cur, dest);
}

__global__ void k_next_level(Arrays A) {
    Ctl& c = *A.ctl;
    c.curCount = c.nextCount;
    c.nextCount = 0; c.binCount = 0; c.anyRedo = 0; c.redoCount = 0; c.sortCount = 0; c.sortMax = 0;
}

// Node record + leaf test + axis choice, main.cu:139-168.
__global__ void k_classify(Arrays A, WorkNode* work, int leafMax) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= A.ctl->curCount) return;
    WorkNode nd = work[w];
    OutNode& o = A.out[nd.out];
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = funkey(nd.lo[k]); mx[k] = funkey(nd.hi[k]); o.lo[k] = mn[k]; o.hi[k] = mx[k]; }
    o.sub = -1; o.height = 1; o.subInternal = 0;
    int n = nd.end - nd.start;
    if (n <= leafMax) {
        o.first = nd.start; o.count = n; o.left = o.right = -1;
        work[w].state = ST_LEAF;
        atomicMax(&A.ctl->largestLeaf, n);
        return;
    }
    if (n <= kSmallNode) { work[w].state = ST_SMALL; return; }      // k_subtree finishes the whole subtree
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    int axis = (dy > dx && dy > dz) ? 1 : ((dz > dx && dz > dy) ? 2 : 0);
    int b = atomicAdd(&A.ctl->binCount, 1);
    work[w].axis = axis; work[w].state = ST_SPLITTING; work[w].bin = b;
    Bins& B = A.bins[b];
    const unsigned kmax = fkey(FLT_MAX), kmin = fkey(-FLT_MAX);      // main.cu:71-75
    for (int i = 0; i < NB; i++) { for (int k = 0; k < 3; k++) { B.lo[i][k] = kmax; B.hi[i][k] = kmin; } B.cnt[i] = 0; }
}

// Bucket fill, main.cu:76-85.
__global__ __launch_bounds__(kBlock) void k_bin(Arrays A, const WorkNode* work, const int* idx, const int* nodeOf, int chunk) {
    __shared__ StageMem<NB> sm;
    stage_reset(sm);
    __syncthreads();
    int cur = -1;
    const BinDest dest{work, A.bins};
    const int b0 = blockIdx.x * chunk, b1 = min(A.n, b0 + chunk);
    for (int base = b0; base < b1; base += kBlock) {
        int p = base + threadIdx.x;
        int w = p < b1 ? nodeOf[p] : -1;
        bool act = w >= 0 && work[w].state == ST_SPLITTING;
        int b = 0;
        bool big = false;
        unsigned kl[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kh[3] = {0u, 0u, 0u};
        if (act) {
            const WorkNode& nd = work[w];
            int id = idx[p], axis = nd.axis;
            big = nd.end - nd.start >= kStageMin;
            float a0 = funkey(nd.lo[axis]), ext = funkey(nd.hi[axis]) - a0;
            float q = NB * (comp3(A.cx, A.cy, A.cz, axis, id) - a0) / ext;
            b = (q == q && fabsf(q) < 1e9f) ? (int)q : 0;           // int(NaN/inf) is UB in the reference; defined 0
            b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
            float4 l = A.lo[id], h = A.hi[id];
            kl[0] = fkey(l.x); kl[1] = fkey(l.y); kl[2] = fkey(l.z); kh[0] = fkey(h.x); kh[1] = fkey(h.y); kh[2] = fkey(h.z);
        }
        direct_add(act && !big, w, b, kl, kh, dest);
        stage_add(sm, cur, act && big, w, b, kl, kh, dest);
    }
    stage_flush(sm, cur, dest);
}

__device__ inline float area_keys(const unsigned lo[3], const unsigned hi[3]) {   // surfaceArea, util.cuh:225-231
    float dx = funkey(hi[0]) - funkey(lo[0]), dy = funkey(hi[1]) - funkey(lo[1]), dz = funkey(hi[2]) - funkey(lo[2]);
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}

// Cost sweep, main.cu:87-131. Right side: bucket i is counted twice (:102-109), kept. Returns the best
// bucket boundary or -1.
__device__ inline int sah_sweep(const Bins& B, const unsigned nlo[3], const unsigned nhi[3]) {
    unsigned rlo[NB][3], rhi[NB][3]; int rc[NB];
    int suffix = B.cnt[NB - 1];
    for (int k = 0; k < 3; k++) { rlo[NB - 1][k] = B.lo[NB - 1][k]; rhi[NB - 1][k] = B.hi[NB - 1][k]; }
    rc[NB - 1] = B.cnt[NB - 1] + suffix;
    for (int i = NB - 2; i >= 1; i--) {
        for (int k = 0; k < 3; k++) { rlo[i][k] = umin_(B.lo[i][k], rlo[i + 1][k]); rhi[i][k] = umax_(B.hi[i][k], rhi[i + 1][k]); }
        suffix += B.cnt[i];
        rc[i] = B.cnt[i] + suffix;
    }
    const float whole = area_keys(nlo, nhi);
    unsigned llo[3], lhi[3]; int lc = B.cnt[0];
    for (int k = 0; k < 3; k++) { llo[k] = B.lo[0][k]; lhi[k] = B.hi[0][k]; }
    float best = FLT_MAX; int bestI = -1;
    for (int i = 1; i < NB; i++) {
        float cost = 1.0f + (lc * area_keys(llo, lhi) + rc[i] * area_keys(rlo[i], rhi[i])) / whole;
        if (cost < best && (lc > 0 && rc[i] > 0)) { best = cost; bestI = i; }
        for (int k = 0; k < 3; k++) { llo[k] = umin_(llo[k], B.lo[i][k]); lhi[k] = umax_(lhi[k], B.hi[i][k]); }
        lc += B.cnt[i];
    }
    return bestI;
}
__global__ void k_sah(Arrays A, WorkNode* work) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= A.ctl->curCount) return;
    if (work[w].state != ST_SPLITTING) return;
    WorkNode nd = work[w];
    const int bestI = sah_sweep(A.bins[nd.bin], nd.lo, nd.hi);
    if (bestI < 0) {
        work[w].state = ST_SORT;
        A.sortRec[atomicAdd(&A.ctl->sortCount, 1)] = SortRec{w, nd.start, nd.end, nd.axis};
        atomicMax(&A.ctl->sortMax, nd.end - nd.start);
        atomicAdd(&A.ctl->sortFallbacks, 1);
    } else {
        float a0 = funkey(nd.lo[nd.axis]), ext = funkey(nd.hi[nd.axis]) - a0;
        work[w].split = a0 + ext * (float(bestI) / float(NB));
    }
}

// ---- small subtrees: one wave each ----------------------------------------------------------------
// A node of at most 64 primitives is finished, down to its leaves, by ONE WAVE holding one primitive per lane: the
// reference's recursion (buildBVH, main.cu:133-233) node by node in its own pre-order (explicit stack, right pushed
// before left, as novum_host.cpp does), every step a wave-level operation — bounds and buckets by masked min/max
// reductions on the order-preserving keys, the sweep on the 12 buckets, the Lomuto partition in its closed form on a
// ballot, the median fallback as a rank-by-counting over the lanes, the centroid mean as a serial sum over lanes in
// position order. No atomics on nodes, no launches, no global traffic but the first load and the last store. The
// deep levels of a tree, where nodes hold a handful of primitives, cost the level-synchronous passes the most.
struct SubStack { short s, e, parent, depthLeft; };             // range relative to the subtree; depthLeft = depth * 2 + isLeft
struct SubLds { Bins bins; SubStack stack[65]; };              // per-wave scratch: every lane writes the same values, then reads them back
__device__ inline int select64(unsigned long long m, int k) {      // position of the k-th (0-based) set bit of m
    int pos = 0;
    for (int w = 32; w; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
        const int c = (int)__popcll(low);
        if (k >= c) { k -= c; m >>= w; pos += w; } else m = low;
    }
    return pos;
}
__device__ inline unsigned wmin_if(bool c, unsigned v) { return wave_min(c ? v : 0xffffffffu); }
__device__ inline unsigned wmax_if(bool c, unsigned v) { return wave_max(c ? v : 0u); }
__global__ __launch_bounds__(kBlock) void k_subtree(Arrays A, WorkNode* work, int* idx, int leafMax) {
    __shared__ SubLds lds[kBlock / 64];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int w = blockIdx.x * (kBlock / 64) + wv;
    if (w >= A.ctl->curCount) return;                           // wave-uniform
    if (work[w].state != ST_SMALL) return;
    SubLds& L = lds[wv];
    const int s0 = work[w].start, n0 = work[w].end - s0;
    int base = 0;
    if (lane == 0) base = atomicAdd(&A.ctl->subCount, 2 * n0 - 1);
    SubNode* sub = A.sub + __shfl(base, 0);
    // one primitive per lane (lanes >= n0 idle); the data moves between lanes as the ranges are permuted
    int id = lane < n0 ? idx[s0 + lane] : 0;
    float c3[3] = {0.0f, 0.0f, 0.0f};
    unsigned kl[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kh[3] = {0u, 0u, 0u};
    if (lane < n0) {
        c3[0] = A.cx[id]; c3[1] = A.cy[id]; c3[2] = A.cz[id];
        const float4 l = A.lo[id], h = A.hi[id];
        kl[0] = fkey(l.x); kl[1] = fkey(l.y); kl[2] = fkey(l.z); kh[0] = fkey(h.x); kh[1] = fkey(h.y); kh[2] = fkey(h.z);
    }
    int top = 0, nLocal = 0, nInternal = 0, height = 0, largest = 0, backups = 0, sorts = 0;
    L.stack[0] = SubStack{0, (short)n0, -1, 2};
    top = 1;
    auto permute = [&](int dest) {                              // lane's data goes to lane `dest` (a permutation of the lanes): ds_permute
        const int a = dest << 2;
        id = __builtin_amdgcn_ds_permute(a, id);
        for (int k = 0; k < 3; k++) {
            c3[k] = __int_as_float(__builtin_amdgcn_ds_permute(a, __float_as_int(c3[k])));
            kl[k] = (unsigned)__builtin_amdgcn_ds_permute(a, (int)kl[k]);
            kh[k] = (unsigned)__builtin_amdgcn_ds_permute(a, (int)kh[k]);
        }
    };
    int guard = 0;                                             // every loop below is bounded by construction; the guards turn a logic
    while (top > 0) {                                          // error into ctl->bad = 4 instead of a spinning wave
        if (++guard > 2 * 64 + 2 || top > 64) { if (lane == 0) A.ctl->bad = 4; break; }
        const SubStack j = L.stack[--top];
        const int s = j.s, e = j.e, n = e - s, depth = j.depthLeft >> 1, nodeId = nLocal++;
        const bool in = lane >= s && lane < e;
        if (lane == 0 && j.parent >= 0) { if (j.depthLeft & 1) sub[j.parent].left = nodeId; else sub[j.parent].right = nodeId; }
        unsigned nlo[3], nhi[3];
        for (int k = 0; k < 3; k++) { nlo[k] = wmin_if(in, kl[k]); nhi[k] = wmax_if(in, kh[k]); }
        bool isLeaf = n <= leafMax;
        int mid = s;
        if (!isLeaf) {
            const float mnx = funkey(nlo[0]), mny = funkey(nlo[1]), mnz = funkey(nlo[2]);
            const float dx = funkey(nhi[0]) - mnx, dy = funkey(nhi[1]) - mny, dz = funkey(nhi[2]) - mnz;
            const int axis = (dy > dx && dy > dz) ? 1 : ((dz > dx && dz > dy) ? 2 : 0);
            const float c = axis == 0 ? c3[0] : (axis == 1 ? c3[1] : c3[2]);
            const float a0 = funkey(nlo[axis]), ext = funkey(nhi[axis]) - a0;
            // buckets, main.cu:71-85
            {
                const unsigned kmax = fkey(FLT_MAX), kmin = fkey(-FLT_MAX);
                if (lane < NB * 3) { (&L.bins.lo[0][0])[lane] = kmax; (&L.bins.hi[0][0])[lane] = kmin; }
                if (lane < NB) L.bins.cnt[lane] = 0;
            }
            float q = NB * (c - a0) / ext;
            int b = (q == q && fabsf(q) < 1e9f) ? (int)q : 0;
            b = b < 0 ? 0 : (b > NB - 1 ? NB - 1 : b);
            // one LDS atomic per box word and lane (min / max / add are exact in any order); the wavefront-scope
            // fences keep the initialising stores, the atomics and the sweep's loads in that order for the compiler
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (in) {
                for (int k = 0; k < 3; k++) { atomicMin(&L.bins.lo[b][k], kl[k]); atomicMax(&L.bins.hi[b][k], kh[k]); }
                atomicAdd(&L.bins.cnt[b], 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int bestI = sah_sweep(L.bins, nlo, nhi);      // every lane, same data
            float split;
            if (bestI < 0) {                                    // median fallback: order by (centroid, index)
                sorts++;
                int rank = 0;
                for (int k = s; k < e; k++) {
                    const float ck = __shfl(c, k); const int ik = __shfl(id, k);
                    rank += (ck < c || (!(c < ck) && ik < id)) ? 1 : 0;
                }
                permute(in ? s + rank : lane);
                const float cc = axis == 0 ? c3[0] : (axis == 1 ? c3[1] : c3[2]);
                split = __shfl(cc, (s0 + s + s0 + e) / 2 - s0);
            } else split = a0 + ext * (float(bestI) / float(NB));
            for (int attempt = 0; attempt < 2; attempt++) {
                const float cc = axis == 0 ? c3[0] : (axis == 1 ? c3[1] : c3[2]);
                const bool good = in && cc < split;
                const unsigned long long G = __ballot(good);
                const int nl = (int)__popcll(G);
                if (nl > 0 && nl < n - 1) {                     // partitionPrimitives, main.cu:49-62, closed form (file header)
                    const int rank = (int)__popcll(G & ((1ull << lane) - 1ull));
                    // position of the k-th element below the split = k-th set bit of G: pure ALU, no table
                    int dest = good ? s + rank : lane;
                    if (in && !good) { for (int hops = 0; dest - s < nl && hops < 64; hops++) dest = select64(G, dest - s); }
                    permute(dest);
                    mid = s + nl;
                    break;                                      // (the reference's second count + partition is a no-op)
                } else if (attempt == 0) {                      // centroid-mean retry, main.cu:192-202: serial sum in position order
                    backups++;
                    float sum = 0.0f;
                    for (int k = s; k < e; k++) sum += __shfl(cc, k);
                    split = sum / n;
                } else isLeaf = true;                           // forced, possibly oversize, leaf
            }
        }
        if (lane == 0) {
            SubNode nd;
            for (int k = 0; k < 3; k++) { nd.lo[k] = funkey(nlo[k]); nd.hi[k] = funkey(nhi[k]); }
            nd.left = nd.right = -1;
            if (isLeaf) { nd.first = s0 + s; nd.count = n; } else { nd.first = -1; nd.count = 0; }
            sub[nodeId] = nd;
        }
        if (!isLeaf) {                                          // all lanes, same values
            L.stack[top] = SubStack{(short)mid, (short)e, (short)nodeId, (short)((depth + 1) * 2)};          // right: after the whole left subtree
            L.stack[top + 1] = SubStack{(short)s, (short)mid, (short)nodeId, (short)((depth + 1) * 2 + 1)};
        }
        if (isLeaf) { largest = largest > n ? largest : n; height = height > depth ? height : depth; }
        else { top += 2; nInternal++; }
    }
    if (lane < n0) idx[s0 + lane] = id;
    if (lane == 0) {
        OutNode& o = A.out[work[w].out];
        o.sub = (int)(sub - A.sub); o.size = nLocal; o.height = height; o.subInternal = nInternal;
        o.left = o.right = -1; o.first = -1; o.count = 0;
        atomicMax(&A.ctl->largestLeaf, largest);
        if (backups) atomicAdd(&A.ctl->backups, backups);
        if (sorts) atomicAdd(&A.ctl->sortFallbacks, sorts);
    }
}

// Median fallback, main.cu:119-128: nth_element's permutation is STL-specific; SURVEY App. D fixes
// it as a sort by (centroid[axis], index). Nodes up to kSortSmall: rank by counting.
__global__ void k_sort_rank(Arrays A, const WorkNode* work, const int* idx, const int* nodeOf) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n) return;
    int w = nodeOf[p];
    if (w < 0 || work[w].state != ST_SORT) return;
    const WorkNode& nd = work[w];
    if (nd.end - nd.start > kSortSmall) return;
    int me = idx[p], axis = nd.axis, rank = 0;
    float ca = comp3(A.cx, A.cy, A.cz, axis, me);
    for (int j = nd.start; j < nd.end; j++) {
        int o = idx[j];
        float cb = comp3(A.cx, A.cy, A.cz, axis, o);
        rank += (cb < ca || (!(ca < cb) && o < me)) ? 1 : 0;
    }
    A.tmp[nd.start + rank] = me;
}
__global__ void k_sort_apply(Arrays A, WorkNode* work, int* idx, const int* nodeOf) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n) return;
    int w = nodeOf[p];
    if (w < 0 || work[w].state != ST_SORT) return;
    if (work[w].end - work[w].start > kSortSmall) return;
    int id = A.tmp[p];
    idx[p] = id;
    if (p == (work[w].start + work[w].end) / 2) work[w].split = comp3(A.cx, A.cy, A.cz, work[w].axis, id);
}
// Larger nodes: bitonic network on 64-bit keys (centroid key << 32 | index — the same total order,
// since centroid keys are ordered like the floats and the builder refuses NaN), padded with ~0 to a
// power of two P >= kSortTile. Strides below kSortTile run in LDS.
__global__ void k_sort_load(Arrays A, const int* idx, SortRec r, int P) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    unsigned long long k = ~0ull;
    if (i < r.end - r.start) {
        int id = idx[r.start + i];
        // the host comparator (`ca < cb`, ties by index) holds -0.0f == +0.0f; fkey() alone would order them, so the
        // zero is canonicalised for THIS key (the bounds keys keep -0 < +0, the rule they share with the host's vmin / vmax)
        float c = comp3(A.cx, A.cy, A.cz, r.axis, id);
        if (c == 0.0f) c = 0.0f;
        k = ((unsigned long long)fkey(c) << 32) | (unsigned)id;
    }
    A.keys[i] = k;
}
__global__ void k_bitonic_global(unsigned long long* keys, int k, int j) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per pair
    int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
    unsigned long long a = keys[lo], b = keys[hi];
    bool up = (lo & k) == 0;
    if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
}
__global__ __launch_bounds__(kBlock) void k_bitonic_lds(unsigned long long* keys, int kFrom, int kTo) {
    __shared__ unsigned long long s[kSortTile];
    const int base = blockIdx.x * kSortTile;
    for (int t = threadIdx.x; t < kSortTile; t += kBlock) s[t] = keys[base + t];
    __syncthreads();
    for (int k = kFrom; k <= kTo; k <<= 1) {
        for (int j = min(k >> 1, kSortTile >> 1); j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < kSortTile / 2; i += kBlock) {
                int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                unsigned long long a = s[lo], b = s[hi];
                bool up = ((base + lo) & k) == 0;
                if ((a > b) == up) { s[lo] = b; s[hi] = a; }
            }
            __syncthreads();
        }
        if (k > INT_MAX / 2) break;
    }
    for (int t = threadIdx.x; t < kSortTile; t += kBlock) keys[base + t] = s[t];
}
__global__ void k_sort_store(Arrays A, WorkNode* work, int* idx, SortRec r) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r.end - r.start) return;
    int id = (int)(unsigned)(A.keys[i] & 0xffffffffull);
    idx[r.start + i] = id;
    if (r.start + i == (r.start + r.end) / 2) work[r.w].split = comp3(A.cx, A.cy, A.cz, r.axis, id);
}

// centroid[axis] < splitPos, main.cu:185-188 (first pass) and :206-209 (after the mean retry).
__global__ void k_flag(Arrays A, const WorkNode* work, const int* idx, const int* nodeOf, int redoOnly) {
    if (redoOnly && !A.ctl->anyRedo) return;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n) return;
    int w = nodeOf[p];
    unsigned char f = 0;
    if (w >= 0) {
        int st = work[w].state;
        if (redoOnly) { if (st != ST_REDO) return; }
        else if (st != ST_SPLITTING && st != ST_SORT) st = 0;
        if (st) f = comp3(A.cx, A.cy, A.cz, work[w].axis, idx[p]) < work[w].split ? 1 : 0;
    } else if (redoOnly) return;
    A.F[p] = f;
}

// ---- exclusive prefix sum of F over all positions: S[0..n], S[n] = total ------------------------
__device__ inline int block_exclusive(int v, int* total) {
    __shared__ int waveSum[kBlock / 64];
    int lane = lane_id(), wv = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) waveSum[wv] = inc;
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < kBlock / 64; i++) { int s = waveSum[i]; if (i < wv) off += s; tot += s; }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}
__device__ inline int load_flags(const unsigned char* F, int n, int base, int f[kScanItems]) {
    int sum = 0;
    if (base + kScanItems <= n) {
        unsigned long long v = *(const unsigned long long*)(F + base);
        for (int j = 0; j < kScanItems; j++) { f[j] = (int)((v >> (8 * j)) & 0xff); sum += f[j]; }
    } else {
        for (int j = 0; j < kScanItems; j++) { f[j] = base + j < n ? F[base + j] : 0; sum += f[j]; }
    }
    return sum;
}
__global__ void k_scan1(Arrays A, int gated) {
    if (gated && !A.ctl->anyRedo) return;
    int f[kScanItems];
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int s = load_flags(A.F, A.n, base, f), tot;
    block_exclusive(s, &tot);
    if (threadIdx.x == 0) A.blockSum[blockIdx.x] = tot;
}
__global__ void k_scan2(Arrays A, int M, int gated) {
    if (gated && !A.ctl->anyRedo) return;
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < M; base += kBlock) {
        int i = base + threadIdx.x;
        int v = i < M ? A.blockSum[i] : 0, tot;
        int ex = block_exclusive(v, &tot);
        int c = carry;
        if (i < M) A.blockSum[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) A.S[A.n] = carry;
}
__global__ void k_scan3(Arrays A, int gated) {
    if (gated && !A.ctl->anyRedo) return;
    int f[kScanItems];
    int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    int s = load_flags(A.F, A.n, base, f), tot;
    int run = A.blockSum[blockIdx.x] + block_exclusive(s, &tot);
    for (int j = 0; j < kScanItems; j++) { if (base + j < A.n) A.S[base + j] = run; run += f[j]; }
}

// numLeft test, main.cu:189-190 (pass 0) and :210-221 (pass 1: forced, possibly oversize, leaf).
__global__ void k_check(Arrays A, WorkNode* work, int pass) {
    if (pass == 1 && !A.ctl->anyRedo) return;
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= A.ctl->curCount) return;
    int st = work[w].state;
    if (pass == 0 ? (st != ST_SPLITTING && st != ST_SORT) : st != ST_REDO) return;
    int s = work[w].start, e = work[w].end, n = e - s;
    int nl = A.S[e] - A.S[s];
    if (nl > 0 && nl < n - 1) { work[w].state = ST_OK; work[w].numLeft = nl; return; }
    if (pass == 0) {
        work[w].state = ST_REDO;
        A.redoList[atomicAdd(&A.ctl->redoCount, 1)] = w;
        A.ctl->anyRedo = 1;
        atomicAdd(&A.ctl->backups, 1);
    } else {
        OutNode& o = A.out[work[w].out];
        o.first = s; o.count = n; o.left = o.right = -1;
        work[w].state = ST_LEAF;
        atomicMax(&A.ctl->largestLeaf, n);
    }
}

// Centroid-mean retry, main.cu:192-202: `sum += c` in index order, then sum / primCount.
__global__ void k_mean(Arrays A, WorkNode* work, const int* idx) {
    if (!A.ctl->anyRedo) return;
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nWaves = (gridDim.x * blockDim.x) >> 6, lane = lane_id();
    int count = A.ctl->redoCount;
    for (int r = wave; r < count; r += nWaves) {
        int w = A.redoList[r];
        int s = work[w].start, e = work[w].end, axis = work[w].axis;
        float sum = 0.0f;
        for (int base = s; base < e; base += 64) {
            int p = base + lane;
            float v = p < e ? comp3(A.cx, A.cy, A.cz, axis, idx[p]) : 0.0f;
            int cnt = e - base < 64 ? e - base : 64;
            for (int k = 0; k < cnt; k++) sum += __shfl(v, k);
        }
        if (lane == 0) work[w].split = sum / (e - s);
    }
}

// Children of the nodes that split, main.cu:226-227 (numbered later).
__global__ void k_alloc(Arrays A, WorkNode* work, WorkNode* next) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= A.ctl->curCount) return;
    if (work[w].state != ST_OK) return;
    int c = atomicAdd(&A.ctl->nextCount, 2), o = atomicAdd(&A.ctl->outCount, 2);
    int s = work[w].start, e = work[w].end, mid = s + work[w].numLeft;
    work[w].child = c;
    OutNode& on = A.out[work[w].out];
    on.left = o; on.right = o + 1; on.first = -1; on.count = 0;
    for (int k = 0; k < 2; k++) {
        WorkNode& ch = next[c + k];
        ch.start = k ? mid : s; ch.end = k ? e : mid; ch.out = o + k; ch.axis = 0;
        for (int j = 0; j < 3; j++) { ch.lo[j] = 0xffffffffu; ch.hi[j] = 0u; }
        ch.split = 0.0f; ch.numLeft = 0; ch.state = 0; ch.child = -1; ch.bin = -1;
    }
}

// g[start + k] = position of the node's k-th element below splitPos.
__global__ void k_goodpos(Arrays A, const WorkNode* work, const int* nodeOf) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.n) return;
    int w = nodeOf[p];
    if (w < 0 || work[w].state != ST_OK || !A.F[p]) return;
    int s = work[w].start;
    A.g[s + (A.S[p] - A.S[s])] = p;
}

// partitionPrimitives, main.cu:49-62, in closed form (header); child bounds, main.cu:139-148.
__global__ __launch_bounds__(kBlock) void k_scatter(Arrays A, const WorkNode* work, WorkNode* next, const int* idx, int* idxOut,
                                                    int* nodeOf, int* nodeOut, int chunk) {
    __shared__ StageMem<2> sm;
    stage_reset(sm);
    __syncthreads();
    int cur = -1;
    const ChildDest dest{work, next};
    const int b0 = blockIdx.x * chunk, b1 = min(A.n, b0 + chunk);
    for (int base = b0; base < b1; base += kBlock) {
        int p = base + threadIdx.x;
        int w = p < b1 ? nodeOf[p] : -1;
        int st = w >= 0 ? work[w].state : 0;
        bool act = st == ST_OK, big = false;
        int side = 0;
        unsigned kl[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, kh[3] = {0u, 0u, 0u};
        if (st == ST_LEAF || st == ST_SMALL) { idxOut[p] = idx[p]; nodeOut[p] = -1; nodeOf[p] = -1; }
        if (act) {
            const WorkNode& nd = work[w];
            int s = nd.start, nl = nd.numLeft, id = idx[p], dst;
            big = nd.end - s >= kStageMin;
            if (A.F[p]) dst = s + (A.S[p] - A.S[s]);
            else {
                dst = p;
                while (dst - s < nl) {
                    int to = A.g[dst];
                    if (to <= dst) { A.ctl->bad = 3; break; }       // cannot happen (header); never spin on the GPU
                    dst = to;
                }
            }
            side = dst < s + nl ? 0 : 1;
            idxOut[dst] = id; nodeOut[dst] = nd.child + side;
            float4 l = A.lo[id], h = A.hi[id];
            kl[0] = fkey(l.x); kl[1] = fkey(l.y); kl[2] = fkey(l.z); kh[0] = fkey(h.x); kh[1] = fkey(h.y); kh[2] = fkey(h.z);
        }
        direct_add(act && !big, w, side, kl, kh, dest);
        stage_add(sm, cur, act && big, w, side, kl, kh, dest);
    }
    stage_flush(sm, cur, dest);
}

// ---- breadth-first -> the reference's depth-first numbering (nodes.size() at push, main.cu:137) ----
__global__ void k_size(Arrays A, int a, int b) {
    int o = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= b) return;
    OutNode& n = A.out[o];
    if (n.sub >= 0) return;                                     // size and height set by k_subtree
    if (n.count > 0) { n.size = 1; n.height = 1; return; }
    const OutNode &l = A.out[n.left], &r = A.out[n.right];
    n.size = 1 + l.size + r.size;
    n.height = 1 + (l.height > r.height ? l.height : r.height);
}
__global__ void k_pre(Arrays A, int a, int b) {
    int o = a + blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= b) return;
    const OutNode& n = A.out[o];
    if (n.count > 0 || n.sub >= 0) return;
    A.out[n.left].pre = n.pre + 1;
    A.out[n.right].pre = n.pre + 1 + A.out[n.left].size;
}
__global__ void k_emit(Arrays A, int total) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const OutNode& n = A.out[o];
    if (n.sub >= 0) {                                           // a subtree: its nodes are already in pre-order
        const SubNode* s = A.sub + n.sub;
        for (int j = 0; j < n.size; j++) {
            pt_bvh_node f;
            f.aabbMIN = pt_float4{s[j].lo[0], s[j].lo[1], s[j].lo[2], 0.0f};
            f.aabbMAX = pt_float4{s[j].hi[0], s[j].hi[1], s[j].hi[2], 0.0f};
            if (s[j].count > 0) { f.left = f.right = -1; f.first = s[j].first; f.primCount = s[j].count; }
            else { f.left = n.pre + s[j].left; f.right = n.pre + s[j].right; f.first = -1; f.primCount = 0; }
            A.fin[n.pre + j] = f;
        }
        return;
    }
    pt_bvh_node f;
    f.aabbMIN = pt_float4{n.lo[0], n.lo[1], n.lo[2], 0.0f};
    f.aabbMAX = pt_float4{n.hi[0], n.hi[1], n.hi[2], 0.0f};
    if (n.count > 0) { f.left = f.right = -1; f.first = n.first; f.primCount = n.count; }
    else { f.left = A.out[n.left].pre; f.right = A.out[n.right].pre; f.first = -1; f.primCount = 0; }
    A.fin[n.pre] = f;
}

inline int blocks(long long n, int per = kBlock) { return (int)std::max<long long>(1, (n + per - 1) / per); }

struct Carver {
    char* base = nullptr; size_t off = 0;
    template <class T> T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

void carve(Carver& c, Arrays& A, int n, int nPos, int maxBins) {
    A.pos = c.take<pt_float4>(nPos); A.mesh = c.take<pt_triangle>(n);
    A.cx = c.take<float>(n); A.cy = c.take<float>(n); A.cz = c.take<float>(n);
    A.lo = c.take<float4>(n); A.hi = c.take<float4>(n);
    A.idxA = c.take<int>(n); A.idxB = c.take<int>(n); A.nodeA = c.take<int>(n); A.nodeB = c.take<int>(n);
    A.tmp = c.take<int>(n); A.S = c.take<int>((size_t)n + 1); A.g = c.take<int>(n);
    A.blockSum = c.take<int>((size_t)blocks(n, kScanTile) + 1);
    A.redoList = c.take<int>(maxBins);
    A.sortRec = c.take<SortRec>(maxBins);
    A.sub = c.take<SubNode>(2 * (size_t)n + 2);
    A.keys = c.take<unsigned long long>(2 * (size_t)n + kSortTile);
    A.F = c.take<unsigned char>((size_t)n + 16);
    A.workA = c.take<WorkNode>((size_t)n + 2); A.workB = c.take<WorkNode>((size_t)n + 2);
    A.bins = c.take<Bins>(maxBins);
    A.out = c.take<OutNode>(2 * (size_t)n); A.fin = c.take<pt_bvh_node>(2 * (size_t)n);
    A.ctl = c.take<Ctl>(1);
}

// ---- re-layout on the device: the packed records the kernels traverse, straight from the
// builder's arrays. Same contents as pt_api.hip's host re-pack; PNodes are numbered in the builder's
// breadth-first order (children of one level in allocation order, which is any valid breadth-first order).
struct PackIn { const pt_float4* normals; int nNormals; const pt_float2* uvs; int nUvs; const int* matType; int nMats; int nLights; };
// mode 0: F = record is an internal node of the breadth-first part; mode 1: F = internal nodes inside the record's subtree
__global__ void k_pack_flags(Arrays A, int total, unsigned char* F, int mode) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const OutNode& n = A.out[o];
    F[o] = mode == 0 ? ((n.count > 0 || n.sub >= 0) ? 0 : 1) : (n.sub >= 0 ? (unsigned char)n.subInternal : 0);
}
// Reference of record c as a child: leaf -> ~first; breadth-first internal -> its number; subtree -> its root's number
// (the subtrees' internal nodes are numbered after all breadth-first ones: nBfs + base of the subtree + local number).
__device__ inline int pack_ref(const Arrays& A, int c, const int* iid, const int* subBase, int nBfs) {
    const OutNode& n = A.out[c];
    if (n.sub >= 0) { const SubNode& r = A.sub[n.sub]; return r.count > 0 ? ~r.first : nBfs + subBase[c]; }
    return n.count > 0 ? ~n.first : iid[c];
}
__global__ void k_pack_nodes(Arrays A, int total, const int* iid, const int* subBase, pt::PNode* nodes, unsigned char* leafEnd) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int nBfs = iid[total];
    const OutNode& n = A.out[o];
    if (n.sub >= 0) {                                           // a subtree: walk its pre-order list, internal nodes numbered in that order
        const SubNode* s = A.sub + n.sub;
        const int base = nBfs + subBase[o];
        int local[127];
        int k = 0;
        for (int j = 0; j < n.size; j++) local[j] = s[j].count > 0 ? -1 : k++;
        for (int j = 0; j < n.size; j++) {
            if (s[j].count > 0) { leafEnd[s[j].first + s[j].count - 1] = 1; continue; }
            const SubNode &Lc = s[s[j].left], &Rc = s[s[j].right];
            pt::PNode p;
            for (int q = 0; q < 3; q++) { p.lmin[q] = Lc.lo[q]; p.lmax[q] = Lc.hi[q]; p.rmin[q] = Rc.lo[q]; p.rmax[q] = Rc.hi[q]; }
            p.left = Lc.count > 0 ? ~Lc.first : base + local[s[j].left];
            p.right = Rc.count > 0 ? ~Rc.first : base + local[s[j].right];
            p.pad0 = p.pad1 = 0;
            nodes[base + local[j]] = p;
        }
        return;
    }
    if (n.count > 0) { leafEnd[n.first + n.count - 1] = 1; return; }
    const OutNode &Lc = A.out[n.left], &Rc = A.out[n.right];
    pt::PNode p;
    for (int k = 0; k < 3; k++) { p.lmin[k] = Lc.lo[k]; p.lmax[k] = Lc.hi[k]; p.rmin[k] = Rc.lo[k]; p.rmax[k] = Rc.hi[k]; }
    p.left = pack_ref(A, n.left, iid, subBase, nBfs);
    p.right = pack_ref(A, n.right, iid, subBase, nBfs);
    p.pad0 = p.pad1 = 0;
    nodes[iid[o]] = p;
}
__global__ void k_pack_tris(Arrays A, const int* idx, PackIn in, const unsigned char* leafEnd, pt::PTri* tris, int* flagsOut) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const int id = idx[i];
    const pt_triangle t = A.mesh[id];
    const pt_float4 a = A.pos[t.aInd], b = A.pos[t.bInd], c = A.pos[t.cInd];      // indices checked by k_prims
    pt::PTri p;
    p.v0[0] = a.x; p.v0[1] = a.y; p.v0[2] = a.z;
    p.e1[0] = b.x - a.x; p.e1[1] = b.y - a.y; p.e1[2] = b.z - a.z;      // trib - tria, integratorUtilities.cuh:13
    p.e2[0] = c.x - a.x; p.e2[1] = c.y - a.y; p.e2[2] = c.z - a.z;      // tric - tria, :14
    p.idx = (uint32_t)id | (leafEnd[i] ? 0x80000000u : 0u);
    int mat = t.materialID;
    if (mat < 0 || mat >= in.nMats) { atomicOr(flagsOut, 1); mat = 0; }
    p.material = mat;
    const int ty = in.matType[mat];
    p.flags = ty == PT_MAT_LEAF ? 1u : 0u;
    if (!(ty == PT_MAT_DIFFUSE || ty == PT_MAT_METAL || ty == PT_MAT_SMOOTHDIELECTRIC || ty == PT_MAT_LEAF || ty == PT_MAT_DELTAMIRROR)) atomicOr(flagsOut, 0x100);
    tris[i] = p;
}
__global__ void k_pack_attrs(Arrays A, PackIn in, pt::PAttr* attrs, int* flagsOut) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const pt_triangle t = A.mesh[i];
    pt::PAttr a;
    const int ni[3] = {t.naInd, t.nbInd, t.ncInd}, ui[3] = {t.uvaInd, t.uvbInd, t.uvcInd};
    float* nd[3] = {a.n0, a.n1, a.n2}; float* ud[3] = {a.uv0, a.uv1, a.uv2};
    for (int k = 0; k < 3; k++) {
        int nk = ni[k], uk = ui[k];
        if (nk < 0 || nk >= in.nNormals) { atomicOr(flagsOut, 2); nk = 0; }
        if (uk < 0 || uk >= in.nUvs) { atomicOr(flagsOut, 4); uk = 0; }
        const pt_float4 nn = in.nNormals > 0 ? in.normals[nk] : pt_float4{0, 0, 0, 0};
        const pt_float2 uu = in.nUvs > 0 ? in.uvs[uk] : pt_float2{0, 0};
        nd[k][0] = nn.x; nd[k][1] = nn.y; nd[k][2] = nn.z;
        ud[k][0] = uu.x; ud[k][1] = uu.y;
    }
    a.emission[0] = t.emission.x; a.emission[1] = t.emission.y; a.emission[2] = t.emission.z;
    a.material = t.materialID;
    a.lightInd = (t.lightInd >= 0 && t.lightInd < in.nLights) ? t.lightInd : -51;
    attrs[i] = a;
}

struct Built {                        // what build_core leaves on the device (pool still allocated)
    void* pool = nullptr;
    Arrays A{};
    int* idx = nullptr;               // the final BVHindices permutation
    int total = 0, outTotal = 0, levels = 0, height = 0;   // reference nodes, breadth-first records, levels built level by level, tree height
    Ctl ctl{};
    hipEvent_t ev0 = nullptr;
    char* extra = nullptr;            // caller's scratch inside the pool
};

#define BVH_HIP(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char m_[384];                                                                         \
            snprintf(m_, sizeof(m_), "pt_bvh_build_device: %s failed: %s", #expr, hipGetErrorString(e_)); \
            if (B.pool) { (void)hipFree(B.pool); B.pool = nullptr; }                              \
            return pt_fail_(-2, m_);                                                              \
        }                                                                                         \
    } while (0)

// Upload, build, number; leaves everything in B (fin nodes in B.A.fin, BVHindices in B.idx). Returns 0 or < 0.
int build_core(const pt_float4* positions, int n_positions, const pt_triangle* triangles, int n_triangles, int max_leaf_size,
               size_t extraBytes, Built& B) {
    if (!positions || !triangles) return pt_fail_(-1, "pt_bvh_build_device: null argument");
    if (n_triangles <= 0 || n_positions <= 0) return pt_fail_(-1, "pt_bvh_build_device: empty scene (the reference aborts with 'No triangles loaded', main.cu:505-508)");
    if (n_triangles > (1 << 28)) return pt_fail_(-1, "pt_bvh_build_device: more than 2^28 triangles");
    if (max_leaf_size < 0) return pt_fail_(-1, "pt_bvh_build_device: negative leaf size");
    const int n = n_triangles;
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) {
            char m[256];
            snprintf(m, sizeof(m), "pt_bvh_build_device: no usable HIP device (%s); the device builder has no CPU fallback (novum_bvh_build_host is the host builder)", hipGetErrorString(e));
            return pt_fail_(-2, m);
        }
    }
    const int maxBins = n / (max_leaf_size + 1) + 2;
    Arrays& A = B.A;
    Carver sizer; carve(sizer, A, n, n_positions, maxBins);
    const size_t extraOff = (sizer.off + 255) & ~(size_t)255;
    BVH_HIP(hipMalloc(&B.pool, extraOff + extraBytes + 256));
    Carver real; real.base = (char*)B.pool; carve(real, A, n, n_positions, maxBins);
    B.extra = (char*)B.pool + extraOff;
    A.n = n; A.nPos = n_positions;
    hipStream_t st = nullptr;
    BVH_HIP(hipEventCreate(&B.ev0));
    BVH_HIP(hipMemcpy((void*)A.pos, positions, sizeof(pt_float4) * (size_t)n_positions, hipMemcpyHostToDevice));
    BVH_HIP(hipMemcpy((void*)A.mesh, triangles, sizeof(pt_triangle) * (size_t)n, hipMemcpyHostToDevice));

    Ctl& ctl = B.ctl;
    ctl = Ctl{};
    ctl.nextCount = 1; ctl.outCount = 1;
    BVH_HIP(hipMemcpy(A.ctl, &ctl, sizeof(ctl), hipMemcpyHostToDevice));
    WorkNode root{};
    root.start = 0; root.end = n; root.out = 0; root.child = -1; root.bin = -1;
    for (int k = 0; k < 3; k++) { root.lo[k] = 0xffffffffu; root.hi[k] = 0u; }
    BVH_HIP(hipMemcpy(A.workA, &root, sizeof(root), hipMemcpyHostToDevice));
    const int gN = blocks(n), gScan = blocks(n, kScanTile);
    const int gChunk = std::min(kChunks, gN);
    const int chunk = ((n + gChunk - 1) / gChunk + kBlock - 1) / kBlock * kBlock;      // positions per workgroup, multiple of the tile
    BVH_HIP(hipEventRecord(B.ev0, st));
    hipLaunchKernelGGL(k_prims, dim3(gChunk), dim3(kBlock), 0, st, A, chunk);

    std::vector<int> levelOff{0};         // out ids of level l are [levelOff[l], levelOff[l+1])
    std::vector<SortRec> recs;
    WorkNode *work = A.workA, *next = A.workB;
    int *idx = A.idxA, *idxOut = A.idxB, *nodeOf = A.nodeA, *nodeOut = A.nodeB;
    int levels = 0;
    long long bound = 1;                  // upper bound of the level's node count before the host knows it
    for (;;) {
        if (levels > 4096) { (void)hipFree(B.pool); B.pool = nullptr; return pt_fail_(-4, "pt_bvh_build_device: more than 4096 levels"); }
        const int gB = blocks(std::min<long long>(bound, n));
        hipLaunchKernelGGL(k_next_level, dim3(1), dim3(1), 0, st, A);
        hipLaunchKernelGGL(k_classify, dim3(gB), dim3(kBlock), 0, st, A, work, max_leaf_size);
        if (kSmallNode > 0) hipLaunchKernelGGL(k_subtree, dim3(blocks(std::min<long long>(bound, n), kBlock / 64)), dim3(kBlock), 0, st, A, work, idx, max_leaf_size);
        hipLaunchKernelGGL(k_bin, dim3(gChunk), dim3(kBlock), 0, st, A, work, idx, nodeOf, chunk);
        hipLaunchKernelGGL(k_sah, dim3(gB), dim3(kBlock), 0, st, A, work);
        BVH_HIP(hipMemcpyAsync(&ctl, A.ctl, sizeof(ctl), hipMemcpyDeviceToHost, st));
        BVH_HIP(hipStreamSynchronize(st));
        if (ctl.bad) {
            (void)hipFree(B.pool); B.pool = nullptr;
            return pt_fail_(ctl.bad >= 3 ? -4 : -1, ctl.bad == 1 ? "pt_bvh_build_device: a triangle's vertex index is out of range"
                                          : ctl.bad == 2 ? "pt_bvh_build_device: non-finite vertex position (the reference's tree is undefined for it)"
                                          : ctl.bad == 4 ? "pt_bvh_build_device: internal error in the small-subtree pass"
                                                         : "pt_bvh_build_device: internal error, partition chain did not advance");
        }
        const int W = ctl.curCount;
        if (W == 0) break;
        levels++;
        levelOff.push_back(ctl.outCount);
        const int gW = blocks(W);
        if (ctl.sortCount > 0) {
            if (ctl.sortMax > kSortSmall) {
                recs.resize(ctl.sortCount);
                BVH_HIP(hipMemcpy(recs.data(), A.sortRec, sizeof(SortRec) * recs.size(), hipMemcpyDeviceToHost));
                for (const SortRec& r : recs) {
                    const int m = r.end - r.start;
                    if (m <= kSortSmall) continue;
                    int P = kSortTile;
                    while (P < m) P <<= 1;
                    hipLaunchKernelGGL(k_sort_load, dim3(blocks(P)), dim3(kBlock), 0, st, A, idx, r, P);
                    hipLaunchKernelGGL(k_bitonic_lds, dim3(P / kSortTile), dim3(kBlock), 0, st, A.keys, 2, kSortTile);
                    for (int k = 2 * kSortTile; k <= P; k <<= 1) {
                        for (int j = k >> 1; j >= kSortTile; j >>= 1)
                            hipLaunchKernelGGL(k_bitonic_global, dim3(P / 2 / kBlock), dim3(kBlock), 0, st, A.keys, k, j);
                        hipLaunchKernelGGL(k_bitonic_lds, dim3(P / kSortTile), dim3(kBlock), 0, st, A.keys, k, k);
                    }
                    hipLaunchKernelGGL(k_sort_store, dim3(blocks(m)), dim3(kBlock), 0, st, A, work, idx, r);
                }
            }
            hipLaunchKernelGGL(k_sort_rank, dim3(gN), dim3(kBlock), 0, st, A, work, idx, nodeOf);
            hipLaunchKernelGGL(k_sort_apply, dim3(gN), dim3(kBlock), 0, st, A, work, idx, nodeOf);
        }
        for (int pass = 0; pass < 2; pass++) {
            hipLaunchKernelGGL(k_flag, dim3(gN), dim3(kBlock), 0, st, A, work, idx, nodeOf, pass);
            hipLaunchKernelGGL(k_scan1, dim3(gScan), dim3(kBlock), 0, st, A, pass);
            hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kBlock), 0, st, A, gScan, pass);
            hipLaunchKernelGGL(k_scan3, dim3(gScan), dim3(kBlock), 0, st, A, pass);
            hipLaunchKernelGGL(k_check, dim3(gW), dim3(kBlock), 0, st, A, work, pass);
            if (pass == 0) hipLaunchKernelGGL(k_mean, dim3(std::min(gW, 2048)), dim3(kBlock), 0, st, A, work, idx);
        }
        hipLaunchKernelGGL(k_alloc, dim3(gW), dim3(kBlock), 0, st, A, work, next);
        hipLaunchKernelGGL(k_goodpos, dim3(gN), dim3(kBlock), 0, st, A, work, nodeOf);
        hipLaunchKernelGGL(k_scatter, dim3(gChunk), dim3(kBlock), 0, st, A, work, next, idx, idxOut, nodeOf, nodeOut, chunk);
        std::swap(work, next); std::swap(idx, idxOut); std::swap(nodeOf, nodeOut);
        bound = 2LL * W;
    }
    // after the last level both index buffers agree on every position (leaves copy, splits scatter)
    const int outTotal = ctl.outCount;                             // breadth-first records (a small subtree is ONE record)
    const int L = (int)levelOff.size() - 1;
    for (int l = L - 1; l >= 0; l--)
        if (levelOff[l + 1] > levelOff[l]) hipLaunchKernelGGL(k_size, dim3(blocks(levelOff[l + 1] - levelOff[l])), dim3(kBlock), 0, st, A, levelOff[l], levelOff[l + 1]);
    BVH_HIP(hipMemsetAsync(&A.out[0].pre, 0, sizeof(int), st));
    for (int l = 0; l < L; l++)
        if (levelOff[l + 1] > levelOff[l]) hipLaunchKernelGGL(k_pre, dim3(blocks(levelOff[l + 1] - levelOff[l])), dim3(kBlock), 0, st, A, levelOff[l], levelOff[l + 1]);
    hipLaunchKernelGGL(k_emit, dim3(blocks(outTotal)), dim3(kBlock), 0, st, A, outTotal);
    OutNode rootRec{};
    BVH_HIP(hipMemcpyAsync(&rootRec, A.out, sizeof(OutNode), hipMemcpyDeviceToHost, st));
    BVH_HIP(hipMemcpyAsync(&ctl, A.ctl, sizeof(ctl), hipMemcpyDeviceToHost, st));     // counters the subtree pass added to
    BVH_HIP(hipStreamSynchronize(st));
    const int total = rootRec.size;
    B.outTotal = outTotal; B.height = rootRec.height;
    B.idx = idx; B.total = total; B.levels = levels;
    return 0;
}

void fill_stats(pt_bvh_build_stats* stats, const Built& B, float ms, std::chrono::steady_clock::time_point wall0) {
    if (!stats) return;
    stats->n_nodes = B.total; stats->largest_leaf = B.ctl.largestLeaf; stats->backups = B.ctl.backups; stats->depth = B.height;
    stats->sort_fallbacks = B.ctl.sortFallbacks; stats->levels = B.levels; stats->device_ms = ms;
    stats->total_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
}

}  // namespace

extern "C" int pt_bvh_build_device(const pt_float4* positions, int n_positions, const pt_triangle* triangles, int n_triangles,
                                   int max_leaf_size, int mode, pt_bvh_node* nodes_out, int nodes_capacity,
                                   int32_t* indices_out, pt_bvh_build_stats* stats) {
    if (mode != PT_BVH_REFERENCE_TREE) return pt_fail_(-3, "pt_bvh_build_device: only PT_BVH_REFERENCE_TREE (0) is built");
    if (!nodes_out || !indices_out) return pt_fail_(-1, "pt_bvh_build_device: null argument");
    const auto wall0 = std::chrono::steady_clock::now();
    Built B;
    if (int r = build_core(positions, n_positions, triangles, n_triangles, max_leaf_size, 0, B)) return r;
    if (B.total > nodes_capacity) { (void)hipFree(B.pool); return pt_fail_(-1, "pt_bvh_build_device: nodes_out too small (2*n_triangles-1 always suffices)"); }
    hipEvent_t ev1 = nullptr;
    BVH_HIP(hipEventCreate(&ev1));
    BVH_HIP(hipEventRecord(ev1, nullptr));
    BVH_HIP(hipMemcpy(nodes_out, B.A.fin, sizeof(pt_bvh_node) * (size_t)B.total, hipMemcpyDeviceToHost));
    BVH_HIP(hipMemcpy(indices_out, B.idx, sizeof(int32_t) * (size_t)n_triangles, hipMemcpyDeviceToHost));
    float ms = 0.0f;
    BVH_HIP(hipEventElapsedTime(&ms, B.ev0, ev1));
    (void)hipEventDestroy(B.ev0); (void)hipEventDestroy(ev1);
    (void)hipFree(B.pool);
    fill_stats(stats, B, ms, wall0);
    return B.total;
}

// Build + re-layout without leaving the device (pt_scene_create_from_mesh, pt_api.hip). d_nodes / d_tris / d_attrs:
// device buffers for n_triangles PNodes, n_triangles PTris, n_triangles PAttrs. out[0..4] = internal nodes, stack need
// (internal nodes on the longest root-to-leaf path), root reference, 1 if a triangle uses a material type without a
// dispatch arm, total reference nodes.
extern "C" int pt_bvh_build_pack_(const pt_scene_desc* d, const int* mat_types, int max_leaf_size,
                                  void* d_nodes, void* d_tris, void* d_attrs, int* out5, pt_bvh_build_stats* stats) {
    const auto wall0 = std::chrono::steady_clock::now();
    const int n = d->n_triangles;
    const size_t szNormals = sizeof(pt_float4) * (size_t)std::max(d->n_normals, 1), szUvs = sizeof(pt_float2) * (size_t)std::max(d->n_uvs, 1);
    const size_t szTypes = sizeof(int) * (size_t)std::max(d->n_materials, 1);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t offUvs = al(szNormals), offTypes = offUvs + al(szUvs), offLeafEnd = offTypes + al(szTypes);
    const size_t offF = offLeafEnd + al((size_t)n + 16), offS = offF + al(2 * (size_t)n + 16), offBlk = offS + al(sizeof(int) * (2 * (size_t)n + 2));
    const size_t offS2 = offBlk + al(sizeof(int) * ((size_t)blocks(2LL * n, kScanTile) + 2));
    const size_t offFlags = offS2 + al(sizeof(int) * (2 * (size_t)n + 2)), extraBytes = offFlags + 256;
    Built B;
    if (int r = build_core(d->positions, d->n_positions, d->triangles, n, max_leaf_size, extraBytes, B)) return r;
    char* X = B.extra;
    if (d->n_normals > 0) BVH_HIP(hipMemcpy(X, d->normals, sizeof(pt_float4) * (size_t)d->n_normals, hipMemcpyHostToDevice));
    if (d->n_uvs > 0) BVH_HIP(hipMemcpy(X + offUvs, d->uvs, sizeof(pt_float2) * (size_t)d->n_uvs, hipMemcpyHostToDevice));
    BVH_HIP(hipMemcpy(X + offTypes, mat_types, sizeof(int) * (size_t)d->n_materials, hipMemcpyHostToDevice));
    BVH_HIP(hipMemsetAsync(X + offLeafEnd, 0, (size_t)n + 16, nullptr));
    BVH_HIP(hipMemsetAsync(X + offFlags, 0, 16, nullptr));
    PackIn in{(const pt_float4*)X, d->n_normals, (const pt_float2*)(X + offUvs), d->n_uvs, (const int*)(X + offTypes), d->n_materials, d->n_lights};
    unsigned char* leafEnd = (unsigned char*)(X + offLeafEnd);
    int* flags = (int*)(X + offFlags);
    // internal nodes of the breadth-first part in that order (exclusive scan of "is internal" over the builder's
    // records), then the subtrees' internal nodes (exclusive scan of their counts)
    const int T = B.outTotal;
    Arrays S = B.A;                                      // the scan kernels read F / S / blockSum / n from their Arrays
    S.F = (unsigned char*)(X + offF); S.S = (int*)(X + offS); S.blockSum = (int*)(X + offBlk); S.n = T;
    Arrays S2 = S;
    S2.S = (int*)(X + offS2);
    hipStream_t st = nullptr;
    const int gT = blocks(T), gScan = blocks(T, kScanTile);
    hipLaunchKernelGGL(k_pack_flags, dim3(gT), dim3(kBlock), 0, st, B.A, T, S.F, 0);
    hipLaunchKernelGGL(k_scan1, dim3(gScan), dim3(kBlock), 0, st, S, 0);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kBlock), 0, st, S, gScan, 0);
    hipLaunchKernelGGL(k_scan3, dim3(gScan), dim3(kBlock), 0, st, S, 0);
    hipLaunchKernelGGL(k_pack_flags, dim3(gT), dim3(kBlock), 0, st, B.A, T, S.F, 1);
    hipLaunchKernelGGL(k_scan1, dim3(gScan), dim3(kBlock), 0, st, S2, 0);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kBlock), 0, st, S2, gScan, 0);
    hipLaunchKernelGGL(k_scan3, dim3(gScan), dim3(kBlock), 0, st, S2, 0);
    hipLaunchKernelGGL(k_pack_nodes, dim3(gT), dim3(kBlock), 0, st, B.A, T, S.S, S2.S, (pt::PNode*)d_nodes, leafEnd);
    hipLaunchKernelGGL(k_pack_tris, dim3(blocks(n)), dim3(kBlock), 0, st, B.A, B.idx, in, leafEnd, (pt::PTri*)d_tris, flags);
    hipLaunchKernelGGL(k_pack_attrs, dim3(blocks(n)), dim3(kBlock), 0, st, B.A, in, (pt::PAttr*)d_attrs, flags);
    hipEvent_t ev1 = nullptr;
    BVH_HIP(hipEventCreate(&ev1));
    BVH_HIP(hipEventRecord(ev1, st));
    int nBfs = 0, nSub = 0, fl = 0;
    OutNode rootRec{};
    SubNode rootSub{};
    BVH_HIP(hipMemcpy(&nBfs, S.S + T, sizeof(int), hipMemcpyDeviceToHost));
    BVH_HIP(hipMemcpy(&nSub, S2.S + T, sizeof(int), hipMemcpyDeviceToHost));
    BVH_HIP(hipMemcpy(&fl, flags, sizeof(int), hipMemcpyDeviceToHost));
    BVH_HIP(hipMemcpy(&rootRec, B.A.out, sizeof(OutNode), hipMemcpyDeviceToHost));
    if (rootRec.sub >= 0) BVH_HIP(hipMemcpy(&rootSub, B.A.sub + rootRec.sub, sizeof(SubNode), hipMemcpyDeviceToHost));
    const int nInternal = nBfs + nSub;
    const int rootRef = rootRec.sub >= 0 ? (rootSub.count > 0 ? ~rootSub.first : nBfs) : (rootRec.count > 0 ? ~rootRec.first : 0);
    float ms = 0.0f;
    BVH_HIP(hipEventElapsedTime(&ms, B.ev0, ev1));
    (void)hipEventDestroy(B.ev0); (void)hipEventDestroy(ev1);
    (void)hipFree(B.pool); B.pool = nullptr;
    if (fl & 1) return pt_fail_(-1, "pt_scene_create_from_mesh: a triangle's material index is out of range");
    if (fl & 2) return pt_fail_(-1, "pt_scene_create_from_mesh: a triangle's normal index is out of range (faces without vn must be given a normal by the loader)");
    if (fl & 4) return pt_fail_(-1, "pt_scene_create_from_mesh: a triangle's uv index is out of range");
    out5[0] = nInternal; out5[1] = B.height - 1; out5[2] = rootRef; out5[3] = (fl & 0x100) ? 1 : 0; out5[4] = B.total;
    fill_stats(stats, B, ms, wall0);
    return 0;
}
