// CHAINED LAUNCH: a run of consecutive convs of one stage (Darknet-53's residual blocks at 76 x 76 and 38 x 38: 1x1, 3x3 + shortcut, 1x1, ...)
// as ONE launch whose resident workgroups take tiles of ALL its layers from a work list in dependency order.
//
// Why (profiles/r05_block_trace.md, profiles/r05_ablation.md): the step's remaining inefficiency sits at the launch boundaries -- the last,
// ragged round of a 3x3 launch runs its workgroups alone on their CUs (38 x 38: 764 tiles on 512 slots = 49 + 45 us where 70 would do), the
// HBM-bound 1x1 launches leave the matrix cores idle and every launch ramps up and drains.  A second stream did not help (half-batch launches
// quantise worse) and splitting the tail inside a launch was slower (memory phases in lockstep).  Here the tiles of layer j + 1 fill the slots
// layer j leaves.
//
// How: the batch is cut into G = 8 groups of consecutive images, one per XCD; group g's tiles (all layers, layer-major) form queue g, served ONLY
// by workgroups that find themselves on XCD g (XCC_ID register), so every producer -> consumer hand-off stays inside one XCD: its L2 is the
// point of coherence, stores need nothing special (they stay in that L2 for the consumer), loads of activations carry sc1 (served by L2, never
// by the CU's L1, which other CUs' stores do not refresh: MI355X_MICROARCH.md, visibility).  Dependencies are per image: a tile of layer j
// waits until every tile of layer j - 1 that touches its images has published (done[g][j-1][image] == need).  Items are claimed in queue order
// and every dependency of an item precedes it in its queue, so whoever holds a claimed item only ever waits for workgroups that already run:
// no deadlock as long as the workgroups of the launch are resident (grid = 2 per CU) -- and a bounded spin that raises a flag instead of
// hanging if that assumption is ever wrong.
//   publish: every wave `s_waitcnt vmcnt(0)` (stores acknowledged by L2), workgroup barrier, one lane adds 1 to done[..] of each image touched;
//   consume: one lane polls done[..] with sc1 loads, workgroup barrier, then the tile (all activation loads sc1).
// The tiles are conv_tap_tile.h / conv_dma_tile.h unchanged (COH instantiations): same arithmetic, same results as the separate launches.
// Buffer reuse across layers stays safe: a layer-(j+1) tile of image i starts after every layer-j tile that reads image i has published, and
// nothing else ever reads image i's rows for a result that is kept (halo positions of neighbouring images feed dropped outputs only).
#include "conv_tap_tile.h"
#include "conv_dma_tile.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef YOLO_CHAIN_SC1       // timing experiment (make EXTRA=-DYOLO_CHAIN_SC1=0): plain activation loads -- NOT coherent, never in the product build
#define YOLO_CHAIN_SC1 1
#endif

namespace yolo {

namespace {
constexpr int kChainTapLds = conv3x3_tap_lds_bytes<2, 4, 4, 26, false>();
constexpr int kChainDmaLds = conv_igemm_dma_lds_bytes<2, 4, 4, 2, 3, 4, false>();
constexpr int kChainLds = kChainTapLds > kChainDmaLds ? kChainTapLds : kChainDmaLds;
constexpr int kChainSpinLimit = 1 << 22;        // x ~0.7 us of s_sleep: seconds, then the flag
// The layer table is read through the CONSTANT address space: written by the host before the launch, never by a kernel -- so that the compiler
// fetches ConvParams fields with scalar loads where they are used, as it does from the kernel arguments of the separate launches (through a plain
// global pointer they became per-lane vector loads held in ~100 VGPRs: 70 spilled).
typedef const ChainLayer __attribute__((address_space(4))) *chain_layer_ptr;
__device__ __forceinline__ ConvParams chain_params(chain_layer_ptr L) {
    ConvParams p;
    __builtin_memcpy(&p, &L->p, sizeof p);
    return p;
}
}  // namespace

// (scalar registers: the tap tile alone needs 91 of the ~100 a wave has, so ONE packed word -- layer << 24 | item -- is all that lives across a
// tile; group, layer record and image range are derived again behind it, through an empty asm statement that keeps the compiler from holding the
// first derivation in registers across the tile)
__device__ __forceinline__ void chain_item_rows(chain_layer_ptr L, int tile, int &i_lo, int &i_hi) {
    const int mt = tile / L->p.n_tiles_n;       // rows [r0, r1) of the layer's tile grid: `unit` per tile, `img_rows` per image
    const int r0 = mt * L->unit;
    const int r1 = r0 + L->unit < L->rows ? r0 + L->unit : L->rows;
    i_lo = r0 / L->img_rows;
    i_hi = (r1 - 1) / L->img_rows;
}

__global__ void __launch_bounds__(512, 4) conv_chain_kernel(const ChainParams cp) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kChainLds];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    for (;;) {
        int state;
        {
            int g = (int)(__builtin_amdgcn_s_getreg(0xF814) & 0xF);         // XCC_ID: the XCD this workgroup runs on
            if (g >= cp.groups) g %= cp.groups;
            if (tid == 0) s_item = __hip_atomic_fetch_add(cp.ctrl + g * kChainCtrlStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int item = __builtin_amdgcn_readfirstlane(s_item);
            __syncthreads();                        // (s_item is rewritten by the next claim)
            if (item >= cp.n_items) break;
#ifdef YOLO_EXPERIMENT
            if (cp.trace && tid == 0) cp.trace[((size_t)g * cp.n_items + item) * 4 + 0] = wall_clock64();
#endif
            int j = 0;
            while (j + 1 < cp.n_layers && item >= cp.first[j + 1]) ++j;
            j = __builtin_amdgcn_readfirstlane(j);
            const int tile = item - cp.first[j];
            state = (j << 24) | tile;
            const chain_layer_ptr L = (chain_layer_ptr)(cp.layers + (size_t)(g * cp.n_layers + j));
            if (j > 0) {
                if (tid == 0) {
                    int i_lo, i_hi;
                    chain_item_rows(L, tile, i_lo, i_hi);
                    const chain_layer_ptr Lp = L - 1;
                    const int *const done = cp.ctrl + kChainDoneOff + (g * cp.n_layers + j - 1) * kChainMaxImages;
                    for (int i = i_lo; i <= i_hi; ++i) {
                        const int need = Lp->need[i];
                        int spins = 0;
                        while (__hip_atomic_load(done + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                            __builtin_amdgcn_s_sleep(20);
                            if (++spins > kChainSpinLimit) {        // never expected: see the header (resident workgroups)
                                __hip_atomic_store(cp.ctrl + kChainTimeoutOff, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                }
                __syncthreads();
            }
#ifdef YOLO_EXPERIMENT
            if (cp.trace && tid == 0) {
                cp.trace[((size_t)g * cp.n_items + item) * 4 + 1] = wall_clock64();
                cp.trace[((size_t)g * cp.n_items + item) * 4 + 3] = __builtin_amdgcn_s_getreg(0xF804);
            }
#endif
        }
        asm volatile("" : "+s"(state));
        {
            int g = (int)(__builtin_amdgcn_s_getreg(0xF814) & 0xF);
            if (g >= cp.groups) g %= cp.groups;
            const int j = state >> 24, tile = state & 0xFFFFFF;
            const chain_layer_ptr L = (chain_layer_ptr)(cp.layers + (size_t)(g * cp.n_layers + j));
            if (L->kind == 0) conv_igemm_dma_tile<2, 4, 4, 2, 3, 4, 4, false, 1, true, YOLO_CHAIN_SC1 != 0>(chain_params(L), smem, tile, tile);
            else conv3x3_tap_tile<false, 2, 4, 4, 4, 26, 4, 1, false, true, false, true, YOLO_CHAIN_SC1 != 0>(chain_params(L), smem, tile, tile, 0, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+s"(state) : : "memory");      // every storing wave: its stores have reached L2
        __syncthreads();                                                    // ... and nobody reads the LDS of this tile any more
        if (tid == 0) {
            int g = (int)(__builtin_amdgcn_s_getreg(0xF814) & 0xF);
            if (g >= cp.groups) g %= cp.groups;
            const int j = state >> 24, tile = state & 0xFFFFFF;
            const chain_layer_ptr L = (chain_layer_ptr)(cp.layers + (size_t)(g * cp.n_layers + j));
            int i_lo, i_hi;
            chain_item_rows(L, tile, i_lo, i_hi);
            int *const done = cp.ctrl + kChainDoneOff + (g * cp.n_layers + j) * kChainMaxImages;
            for (int i = i_lo; i <= i_hi; ++i) __hip_atomic_fetch_add(done + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef YOLO_EXPERIMENT
            if (cp.trace) cp.trace[((size_t)g * cp.n_items + cp.first[j] + tile) * 4 + 2] = wall_clock64();
#endif
        }
    }
    // the last workgroup to leave returns the counters to zero for the next launch
    if (tid == 0) {
        int *const exited = cp.ctrl + kChainExitOff;
        if (__hip_atomic_fetch_add(exited, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            for (int q = 0; q < cp.groups; ++q) __hip_atomic_store(cp.ctrl + q * kChainCtrlStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int nd = cp.groups * cp.n_layers * kChainMaxImages;
            for (int q = 0; q < nd; ++q) __hip_atomic_store(cp.ctrl + kChainDoneOff + q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(exited, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Completes one layer of a chain for a group of `imgs` images: tile counts, grid of the tile kind, the rows-per-image bookkeeping of the
// dependency counters.  p: the layer's ConvParams at batch = imgs (pointers at the group's first image).  False: not chainable.
bool conv_chain_layer(ChainLayer &L, const ConvParams &p0, int kind, int imgs) {
    if (imgs < 1 || imgs > kChainMaxImages) return false;
    L.p = p0;
    L.kind = kind;
    ConvParams &p = L.p;
    if (p.f32 || p.ksplit > 1 || p.fuse2 || !conv_fast_epilogue_ok(p) || p.obj_out || p.M != imgs * p.HoWo) return false;
    p.fast_epi = 1;
    if (kind == 0) {
        if (p.ksize != 1 || p.stride != 1 || p.pad != 0 || (p.cin_chunks & 3) || prepare_conv_dma(p, 14) != hipSuccess) return false;
        L.unit = 128; L.rows = p.M; L.img_rows = p.HoWo;
    } else {
        if (p.ksize != 3 || p.stride != 1 || p.pad != 1 || (p.cin_chunks & 3) || p.Ho != p.H || p.Wo != p.W || !conv_tap_fits(0, p.W) ||
            prepare_conv_dma(p, 8) != hipSuccess || p.q_stride != 256)
            return false;
        L.unit = 256; L.rows = p.Mq; L.img_rows = p.qHW;
    }
    L.tiles = p.n_blocks;
    for (int i = 0; i < kChainMaxImages; ++i) L.need[i] = 0;
    const int mts = p.n_blocks / p.n_tiles_n;
    for (int mt = 0; mt < mts; ++mt) {
        const int r0 = mt * L.unit, r1 = r0 + L.unit < L.rows ? r0 + L.unit : L.rows;
        for (int i = r0 / L.img_rows; i <= (r1 - 1) / L.img_rows; ++i) {
            if (i >= imgs) return false;
            L.need[i] += p.n_tiles_n;
        }
    }
    return true;
}

hipError_t launch_conv_chain(const ChainParams &cp, hipStream_t s) {
    if (!cp.layers || !cp.ctrl || cp.n_layers < 2 || cp.n_layers > kChainMaxLayers || cp.groups < 1 || cp.groups > kChainGroups || cp.n_items < 1)
        return hipErrorInvalidValue;
#ifdef YOLO_EXPERIMENT      // YOLO_CHAIN_TRACE=<file>: per-item timestamps of every chained launch (tools/chain_trace.py); synchronous
    if (const char *tf = getenv("YOLO_CHAIN_TRACE")) {
        ChainParams c = cp;
        const size_t n = (size_t)cp.groups * cp.n_items * 4;
        unsigned long long *dev = nullptr;
        if (hipMalloc((void **)&dev, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(dev, 0, n * 8);
        c.trace = dev;
        hipLaunchKernelGGL(conv_chain_kernel, dim3(512), dim3(512), 0, s, c);
        hipError_t e = hipStreamSynchronize(s);
        std::vector<unsigned long long> host(n);
        if (e == hipSuccess) e = hipMemcpy(host.data(), dev, n * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dev);
        if (e == hipSuccess)
            if (FILE *fp = fopen(tf, "ab")) {
                unsigned long long hdr[4 + kChainMaxLayers + 1] = {(unsigned long long)cp.groups, (unsigned long long)cp.n_items, (unsigned long long)cp.n_layers, 0ull};
                for (int j = 0; j <= kChainMaxLayers; ++j) hdr[4 + j] = (unsigned long long)cp.first[j];
                fwrite(hdr, 8, 4 + kChainMaxLayers + 1, fp);
                fwrite(host.data(), 8, n, fp);
                fclose(fp);
            }
        return e;
    }
#endif
    hipLaunchKernelGGL(conv_chain_kernel, dim3(512), dim3(512), 0, s, cp);
    return hipGetLastError();
}

const char *conv_chain_symbol() { return "yolo::conv_chain_kernel(yolo::ChainParams)"; }

}  // namespace yolo
