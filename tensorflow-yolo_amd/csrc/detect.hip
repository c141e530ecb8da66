// YOLO head decode + greedy NMS on the device.
//
// decode_kernel   replaces net/v2.py:93-119 and net/v3.py:109-136 (the triple Python loop):
//                 one thread per (cell, anchor) row; rows that pass the score threshold are
//                 appended (wavefront-aggregated atomic) to a per-image candidate list together
//                 with their scan index.
// nms_kernel      replaces net/base.py:180-209: one workgroup per image; candidates are sorted in
//                 LDS by (prob descending, scan index ascending) == Python's stable
//                 list.sort(key=prob, reverse=True) over the scan-ordered list (base.py:199),
//                 then suppressed greedily; IoU in float64 exactly as base.py:180-192 computes it
//                 (x,y float32; w,h float64; union floored at 1e-8; suppress when iou >= thr).
#include <atomic>

#include "yolo_internal.h"

namespace yolo {

__device__ __forceinline__ float sigmoid_f32(float x) { return 1.f / (1.f + expf(-x)); }   // base.py:171-172

__global__ void __launch_bounds__(256) decode_kernel(const DecodeParams p) {
    const int width = 5 + p.n_classes;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    // (whole waves stay in the loop together: the v3 path uses wave-wide shuffles)
    for (long long row0 = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); row0 < p.total_rows; row0 += stride) {
        const bool live = row0 + lane < p.total_rows;
        const long long row = live ? row0 + lane : p.total_rows - 1;
        // (32-bit division where the row index allows it: always, up to batch 94 000 of YOLOv3-608)
        const int b = p.total_rows <= 0x7fffffffLL ? (int)((unsigned)row / (unsigned)p.rows) : (int)(row / p.rows);
        const int r = (int)(row - (long long)b * p.rows);
        const float *t = p.logits + row * width;
        const float po = sigmoid_f32(p.obj ? p.obj[row] : t[4]);      // compact copy written by the head convs: coalesced
        float prob;
        int cls = 0;
        if (p.version == 3) {
            prob = po;                                  // v3.py:123 p = prob_obj
            // v3.py:120-121 argmax of sigmoid(cls), first max wins.  The few rows of a wave that pass the threshold
            // (v3.py:124, p == thr is kept) are served by 16-lane groups (five classes per lane + a butterfly argmax
            // that prefers the lower index on ties) instead of 80 serial sigmoids on one lane.
            unsigned long long todo = __ballot(live && !(prob < p.threshold));
            const int grp3 = lane >> 4, gl3 = lane & 15;
            while (todo) {          // four passing rows at a time, one per 16-lane group (as the v2 path below)
                int src[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    src[g] = todo ? __ffsll((long long)todo) - 1 : -1;
                    if (todo) todo &= todo - 1;
                }
                const int mine = grp3 == 0 ? src[0] : grp3 == 1 ? src[1] : grp3 == 2 ? src[2] : src[3];
                const int from = mine < 0 ? lane : mine;
                const long long srow = ((long long)__shfl((int)(row >> 32), from) << 32) | (unsigned)__shfl((int)(row & 0xffffffffLL), from);
                const float *ts = p.logits + srow * width + 5;
                float s0 = -1.f;
                int k0 = 0x7fffffff;
                if (mine >= 0)
                    for (int k = gl3; k < p.n_classes; k += 16) {
                        const float sk = sigmoid_f32(ts[k]);
                        if (sk > s0) { s0 = sk; k0 = k; }  // ascending k per lane: strict > keeps the first maximum
                    }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const float so = __shfl_xor(s0, off);
                    const int ko = __shfl_xor(k0, off);
                    if (so > s0 || (so == s0 && ko < k0)) { s0 = so; k0 = ko; }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kb = __shfl(k0, 16 * g);
                    if (lane == src[g]) cls = kb;
                }
            }
            if (!live || prob < p.threshold) continue;
        } else {
            // v2.py:102-106: p = sigmoid(obj) * max softmax(cls), class = argmax (first maximum).  p <= sigmoid(obj), so only rows whose
            // objectness alone reaches the threshold need the softmax -- and those rows of a wave are served by 16-lane groups
            // (butterfly max / sum / argmax) instead of three serial passes over the classes on one lane while 63 idle (round 4:
            // YOLOv2-416 b16 49 -> 16 us; the sum of the exponentials is a butterfly sum now, not a serial one: the probabilities
            // move by an ulp -- base.py:175-177's own pairwise sum is neither; the reference's goldens hold to 2e-6 either way).
            float best = 0.f;
            unsigned long long todo = __ballot(live && !(po < p.threshold));
            // four rows at a time, one per 16-lane group (80 classes = five per lane; the butterflies stay inside a group): the passing
            // rows of a wave are a serial chain of dependent loads (the row's class logits -> max -> exp -> sum), and at a threshold of
            // 0.5 a third of YOLOv2's rows pass on objectness alone
            const int grp = lane >> 4, gl = lane & 15;
            while (todo) {
                int src[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    src[g] = todo ? __ffsll((long long)todo) - 1 : -1;
                    if (todo) todo &= todo - 1;
                }
                const int mine = grp == 0 ? src[0] : grp == 1 ? src[1] : grp == 2 ? src[2] : src[3];      // the row this lane's group serves
                const int from = mine < 0 ? lane : mine;
                const long long srow = ((long long)__shfl((int)(row >> 32), from) << 32) | (unsigned)__shfl((int)(row & 0xffffffffLL), from);
                const float *ts = p.logits + srow * width + 5;
                float mx = -3.0e38f;
                if (mine >= 0)
                    for (int k = gl; k < p.n_classes; k += 16) mx = fmaxf(mx, ts[k]);
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                float sum = 0.f;
                if (mine >= 0)
                    for (int k = gl; k < p.n_classes; k += 16) sum += expf(ts[k] - mx);
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                float s0 = -1.f;
                int k0 = 0x7fffffff;
                if (mine >= 0)
                    for (int k = gl; k < p.n_classes; k += 16) {
                        const float sk = expf(ts[k] - mx) / sum;
                        if (sk > s0) { s0 = sk; k0 = k; }      // ascending k per lane: strict > keeps the first maximum
                    }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {
                    const float so = __shfl_xor(s0, off);
                    const int ko = __shfl_xor(k0, off);
                    if (so > s0 || (so == s0 && ko < k0)) { s0 = so; k0 = ko; }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {       // each group's verdict back to the lane that owns the row
                    const float sb = __shfl(s0, 16 * g);
                    const int kb = __shfl(k0, 16 * g);
                    if (lane == src[g]) { best = sb; cls = kb; }
                }
            }
            if (!live || po < p.threshold) continue;    // p = po * pc <= po: same result as testing p alone
            prob = po * best;                           // v2.py:106
            if (prob < p.threshold) continue;           // v2.py:107
        }
        int s = 0;
        while (s + 1 < p.n_scales && r >= p.sc[s + 1].row0) ++s;
        const DecodeScale &sc = p.sc[s];
        const int rr = r - sc.row0;
        const int a = rr % sc.na;
        const int cell = rr / sc.na;
        const int cw = cell % sc.w, cy = cell / sc.w;
        Candidate c;
        c.x = (sigmoid_f32(t[0]) + (float)cw) / (float)sc.w;            // v2.py:112 / v3.py:129 (float32)
        c.y = (sigmoid_f32(t[1]) + (float)cy) / (float)sc.h;
        c.w = (sc.aw[a] * (double)expf(t[2])) / (double)sc.w;           // float64: anchors are np.float64
        c.h = (sc.ah[a] * (double)expf(t[3])) / (double)sc.h;
        c.prob = prob;
        c.cls = cls;
        c.scan = (unsigned)r;
        c.pad_ = 0;
        const int slot = atomicAdd(&p.cand_count[b * kCandCountStride], 1);
        if (slot < p.cap) reinterpret_cast<Candidate *>(p.cand)[(long long)b * p.cap + slot] = c;
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned orderable(float f) {        // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// bytes of the union region { sort keys | sorted boxes (+ suppression bit matrix of the <= 512-candidate path) }
__host__ __device__ inline size_t nms_union_bytes(size_t cap2) {
    const size_t u = (cap2 * 29 + 15) & ~(size_t)15;            // max(key 8 B, boxes 8+8+4+4+4+1 = 29 B) per entry
    return u < 49152 ? 49152 : u;                               // boxes of 512 entries (< 16 KiB) + 512 x 8 x 8 B of mask
}

// Working storage carve (16-byte aligned): idx u16[n2] | union { key u64[n2] ; boxes } | kept u16[cap].
// Up to 4096 candidates per image it lives in LDS (dynamic); above (GLOBAL: up to 65536, e.g. a mAP-style
// threshold of 0.005 on 22 743 rows) the same algorithm runs on a per-image slab of global memory that stays in L2
// (one workgroup per image either way: __syncthreads orders the slab accesses inside the CU).
// An image with at most 512 candidates (the usual case) takes the single-wave path, whose working set (carved for a
// capacity of 512: 51 KiB) always lives in LDS -- in the GLOBAL instantiation too, where the single-wave sort would
// otherwise exchange keys between lanes through global memory with nothing but in-order L1 behaviour to order them.
extern __shared__ __attribute__((aligned(16))) unsigned char nms_dyn_lds[];
constexpr int kNmsSmall = 512;
constexpr size_t kNmsSmallLds = 1024 + 49152 + 1024 + 64;        // idx | union | kept for cap2 = 512 (nms_lds_bytes(512))
template <bool GLOBAL>
__global__ void __launch_bounds__(1024) nms_kernel(const NmsParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char small_lds[GLOBAL ? kNmsSmallLds : 16];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int nthr = blockDim.x;
    int count = p.cand_count[b * kCandCountStride];
    int status = 0;
    if (count > p.cap) { count = p.cap; status = 1; }
    const int n = count;
    if (n == 0) {
        if (tid == 0) { p.counts[b] = 0; p.status[b] = status; }
        return;
    }
    // (every thread has read the count: the sort's first barrier lies between this store and anything that follows the kernel)
    const bool small = n <= kNmsSmall;  // the usual case: single-wave sort + suppression bit matrix + single-wave greedy pass
    unsigned char *lds = !GLOBAL ? nms_dyn_lds : small ? small_lds : p.scratch + (size_t)blockIdx.x * p.scratch_stride;
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    int cap2 = 2;
    while (cap2 < ((GLOBAL && small) ? kNmsSmall : p.cap)) cap2 <<= 1;

    unsigned short *idx = reinterpret_cast<unsigned short *>(lds);
    unsigned char *u = lds + (((size_t)cap2 * 2 + 15) & ~(size_t)15);
    unsigned long long *key = reinterpret_cast<unsigned long long *>(u);
    const Candidate *cand = p.cand + (long long)b * p.cap;

    for (int i = tid; i < n2; i += nthr) {
        unsigned long long k = ~0ULL;
        if (i < n) k = ((unsigned long long)(~orderable(cand[i].prob)) << 32) | cand[i].scan;
        key[i] = k;
        idx[i] = (unsigned short)i;
    }
    __syncthreads();
    if (tid == 0 && p.reset_count) p.reset_count[b * kCandCountStride] = 0;       // every thread has read the count (barrier above)
    if (small) {
        // <= 512 candidates: RANK sort -- the keys (prob descending, scan index ascending) are unique, so the place of candidate i in
        // the sorted list is the number of keys below its own: every thread below n counts through the n keys (LDS broadcast reads,
        // no exchange stages, one barrier) and drops its index at its rank.  (Round 4; was a single-wave bitonic network: 28 stages of
        // dependent LDS round trips for 100 candidates, 45 for 512 -- most of the 28 us this kernel took for one YOLOv3-608 image.)
        unsigned short *const sidx = reinterpret_cast<unsigned short *>(key + n2);     // behind the keys (the union region holds 29 B per entry)
        if (tid < n) {
            const unsigned long long mine = key[tid];
            int rank = 0;
            // (ties broken by the slot index: a total order for ANY input -- yolo_net_detect's scan indices are unique, those a caller of
            // yolo_nms_host / yolo_decode_nms passes need not be, and two equal ranks would leave a slot of sidx unwritten)
            for (int j = 0; j < n; ++j) { const unsigned long long kj = key[j]; rank += (kj < mine) || (kj == mine && j < tid); }
            sidx[rank] = (unsigned short)tid;
        }
        __syncthreads();
        if (tid < n) idx[tid] = sidx[tid];
        __syncthreads();
    } else
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += nthr) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long ki = key[i], kl = key[l];
                if ((ki > kl) == up) {
                    key[i] = kl; key[l] = ki;
                    const unsigned short s = idx[i]; idx[i] = idx[l]; idx[l] = s;
                }
            }
            __syncthreads();
        }
    }
    // sorted boxes into LDS (aliases the key array: read idx first, barrier, then overwrite)
    double *bw = reinterpret_cast<double *>(u);
    double *bh = bw + n;
    float *bx = reinterpret_cast<float *>(bh + n);
    float *by = bx + n;
    int *bc = reinterpret_cast<int *>(by + n);
    unsigned char *alive = reinterpret_cast<unsigned char *>(bc + n);
    unsigned short *kept = reinterpret_cast<unsigned short *>(u + nms_union_bytes((size_t)cap2));
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        const Candidate c = cand[idx[i]];
        bw[i] = c.w; bh[i] = c.h; bx[i] = c.x; by[i] = c.y; bc[i] = c.cls; alive[i] = 1;
    }
    __syncthreads();

    const double thr = p.iou_threshold;
    int nk = 0;
    bool truncated = false;
    if (small) {
        // (1) every wave: one 64-bit word at a time of the suppression matrix, mask[i][w] bit jj = box 64 w + jj (later in the
        //     sorted order than i) has IoU >= thr with box i -- the same float64 arithmetic as the loop below;
        // (2) wave 0: greedy pass over the sorted list with the alive set in registers (lane w holds word w):
        //     alive &= ~mask[i] for every kept i.  Five barriers per image instead of one per survivor.
        const int W = (n + 63) >> 6;
        unsigned long long *mask = reinterpret_cast<unsigned long long *>(u + 16384);
        int *scal = reinterpret_cast<int *>(u + 16384 - 16);                // nk, truncated (the boxes of <= 512 entries end below)
        // one WAVE per word: lane jj tests box 64 w + jj against box i, the word is the ballot (a thread per word walked its 64
        // boxes alone -- 180 busy threads of 1024 and ~20 us of float64 divisions in a row for a 90-box image)
        const int lane = tid & 63, nwaves = nthr >> 6;
        for (int item = tid >> 6; item < n * W; item += nwaves) {       // wave-uniform
            const int i = item / W, w = item - i * W;
            const int j = 64 * w + lane;
            bool hit = false;
            if (j > i && j < n && !(p.mode == YOLO_NMS_PER_CLASS && bc[j] != bc[i])) {
                const double w1 = bw[i], h1 = bh[i];
                const double x1 = (double)bx[i], y1 = (double)by[i];
                const double ax1 = (x1 - w1 / 2.) * 1., ay1 = (y1 - h1 / 2.) * 1.;      // base.py:267-272
                const double ax2 = (x1 + w1 / 2.) * 1., ay2 = (y1 + h1 / 2.) * 1.;
                const double a1 = w1 * h1;
                const double w2 = bw[j], h2 = bh[j];
                const double x2 = (double)bx[j], y2 = (double)by[j];
                const double bx1 = (x2 - w2 / 2.) * 1., by1 = (y2 - h2 / 2.) * 1.;
                const double bx2 = (x2 + w2 / 2.) * 1., by2 = (y2 + h2 / 2.) * 1.;
                const double iw = fmax(fmin(ax2, bx2) - fmax(ax1, bx1), 0.);
                const double ih = fmax(fmin(ay2, by2) - fmax(ay1, by1), 0.);
                const double inter = iw * ih;
                const double uni = fmax(a1 + w2 * h2 - inter, 1e-8);                // base.py:190
                hit = inter / uni >= thr;                                           // base.py:204
            }
            const unsigned long long bits = __ballot(hit);
            if (lane == 0) mask[i * 8 + w] = bits;
        }
        __syncthreads();
        if (tid < 64) {
            unsigned long long alive_w = 0;
            if (tid < W) alive_w = (tid == W - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1) : ~0ull;
            for (int i = 0; i < n; ++i) {
                const int wsel = i >> 6;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(alive_w & 0xffffffffull), wsel);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(alive_w >> 32), wsel);
                const unsigned long long wv = ((unsigned long long)hi << 32) | lo;
                if (!((wv >> (i & 63)) & 1)) continue;
                if (nk == p.max_boxes) { truncated = true; break; }
                if (tid == 0) kept[nk] = (unsigned short)i;
                ++nk;
                if (tid < W) alive_w &= ~mask[i * 8 + tid];
            }
            if (tid == 0) { scal[0] = nk; scal[1] = truncated ? 1 : 0; }
        }
        __syncthreads();
        nk = scal[0];
        truncated = scal[1] != 0;
    } else
    for (int i = 0; i < n; ++i) {
        if (!alive[i]) continue;                // final: every earlier survivor's pass ended with a barrier
        if (nk == p.max_boxes) { truncated = true; break; }
        if (tid == 0) kept[nk] = (unsigned short)i;
        ++nk;
        const double w1 = bw[i], h1 = bh[i];
        const double x1 = (double)bx[i], y1 = (double)by[i];
        const double ax1 = (x1 - w1 / 2.) * 1., ay1 = (y1 - h1 / 2.) * 1.;      // base.py:267-272
        const double ax2 = (x1 + w1 / 2.) * 1., ay2 = (y1 + h1 / 2.) * 1.;
        const double a1 = w1 * h1;
        const int c1 = bc[i];
        for (int j = i + 1 + tid; j < n; j += nthr) {
            if (!alive[j]) continue;
            if (p.mode == YOLO_NMS_PER_CLASS && bc[j] != c1) continue;
            const double w2 = bw[j], h2 = bh[j];
            const double x2 = (double)bx[j], y2 = (double)by[j];
            const double bx1 = (x2 - w2 / 2.) * 1., by1 = (y2 - h2 / 2.) * 1.;
            const double bx2 = (x2 + w2 / 2.) * 1., by2 = (y2 + h2 / 2.) * 1.;
            const double iw = fmax(fmin(ax2, bx2) - fmax(ax1, bx1), 0.);
            const double ih = fmax(fmin(ay2, by2) - fmax(ay1, by1), 0.);
            const double inter = iw * ih;
            const double uni = fmax(a1 + w2 * h2 - inter, 1e-8);                // base.py:190
            if (inter / uni >= thr) alive[j] = 0;                               // base.py:204
        }
        __syncthreads();
    }
    __syncthreads();
    if (truncated) status |= 2;
    for (int k = tid; k < nk; k += nthr) {
        const int i = kept[k];
        yolo_box o;
        o.x = bx[i]; o.y = by[i]; o.w = (float)bw[i]; o.h = (float)bh[i];
        o.class_idx = bc[i];
        const int ci = idx[i];
        o.prob = cand[ci].prob;
        p.boxes[(long long)b * p.max_boxes + k] = o;
        if (p.keep_idx) p.keep_idx[(long long)b * p.max_boxes + k] = ci;
    }
    if (tid == 0) { p.counts[b] = nk; p.status[b] = status; }
}

size_t nms_lds_bytes(int cap) {
    size_t cap2 = 2;
    while ((int)cap2 < cap) cap2 <<= 1;
    size_t a = (cap2 * 2 + 15) & ~(size_t)15;           // idx
    return a + nms_union_bytes(cap2) + cap2 * 2 + 64;   // + kept
}

hipError_t launch_decode(const DecodeParams &p, int batch, hipStream_t s, bool zero_counts) {
    if (zero_counts) {      // (not needed behind a detect whose NMS returned the counters to zero: NmsParams.reset_count)
        hipError_t e = hipMemsetAsync(p.cand_count, 0, sizeof(int) * (size_t)batch * kCandCountStride, s);
        if (e != hipSuccess) return e;
    }
    long long g = (p.total_rows + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)g), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_nms(const NmsParams &p, int batch, hipStream_t s) {
    const size_t lds = nms_lds_bytes(p.cap);
    if (lds > 160 * 1024) {         // global-memory slabs
        if (p.cap > 65536 || !p.scratch || p.scratch_stride < lds) return hipErrorInvalidValue;
        hipLaunchKernelGGL(nms_kernel<true>, dim3((unsigned)batch), dim3(1024), 0, s, p);
        return hipGetLastError();
    }
    // the dynamic-LDS limit is a per-device property of the function: remember what was set per device ordinal
    static std::atomic<size_t> configured[64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev < 0 || dev >= 64 || lds > configured[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) configured[dev].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(nms_kernel<false>, dim3((unsigned)batch), dim3(1024), lds, s, p);
    return hipGetLastError();
}

// bytes of global scratch per image the NMS needs for this capacity (0: the LDS path is used)
size_t nms_scratch_bytes(int cap) {
    const size_t lds = nms_lds_bytes(cap);
    return lds > 160 * 1024 ? ((lds + 255) & ~(size_t)255) : 0;
}

}  // namespace yolo
