// Bandwidth kernels around the conv stack: input cast/pad, max-pool, and the generic strided
// element-wise fallback (standalone shortcut / upsample / reorg / concat-copy / f32 convert) the
// planner uses when a fusion into a conv epilogue is not possible.
#include "yolo_internal.h"
#include <type_traits>

namespace yolo {

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// feed point of the graph (net/layers.py:106-109): float32 NHWC -> T NHWC with the channel count
// padded to one 16-byte chunk (zeros), so the first conv can run the chunked implicit GEMM.
template <bool F32>
__global__ void __launch_bounds__(256) prep_kernel(const PrepParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long px = (long long)blockIdx.x * blockDim.x + threadIdx.x; px < p.pixels; px += stride) {
        const float *src = p.in + px * p.C;
        T *dst = reinterpret_cast<T *>(p.out) + px * p.Cpad;
        for (int c0 = 0; c0 < p.Cpad; c0 += 16 / (int)sizeof(T)) {
            T t[16 / sizeof(T)];
#pragma unroll
            for (int e = 0; e < 16 / (int)sizeof(T); ++e) t[e] = (c0 + e < p.C) ? (T)src[c0 + e] : (T)0.f;
            uint4v u;
            __builtin_memcpy(&u, t, 16);
            *reinterpret_cast<uint4v *>(dst + c0) = u;
        }
    }
}

// net/layers.py:70-81.  stride 2: zero pad (0 before, 1 after) then 2x2 VALID -- the pad row/col is
// only read for odd H/W and then takes part in the max as 0.  stride 1: TF SAME, window clipped.
template <bool F32, bool VEC>
__global__ void __launch_bounds__(256) pool_kernel(const PoolParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int STEP = VEC ? EPC : 1;
    const int cchunks = (p.C + STEP - 1) / STEP;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < p.total; w += stride) {
        const int cc = (int)(w % cchunks);
        long long t = w / cchunks;
        const int ox = (int)(t % p.Wo); t /= p.Wo;
        const int oy = (int)(t % p.Ho);
        const long long n = t / p.Ho;
        const int iy = oy * p.stride, ix = ox * p.stride;
        float best[STEP];
#pragma unroll
        for (int e = 0; e < STEP; ++e) best[e] = -INFINITY;
        const T *base = reinterpret_cast<const T *>(p.in) + n * p.in_img_stride + cc * STEP;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = iy + dy, x = ix + dx;
                if (y < p.H && x < p.W) {
                    const T *q = base + ((long long)y * p.W + x) * p.in_ld;
                    if (VEC) {
                        const uint4v u = *reinterpret_cast<const uint4v *>(q);
                        T tv[EPC];
                        __builtin_memcpy(tv, &u, 16);
#pragma unroll
                        for (int e = 0; e < STEP; ++e) best[e] = fmaxf(best[e], (float)tv[e]);
                    } else {
                        best[0] = fmaxf(best[0], (float)q[0]);
                    }
                } else if (p.stride == 2) {     // explicit zero padding takes part (layers.py:72-73)
#pragma unroll
                    for (int e = 0; e < STEP; ++e) best[e] = fmaxf(best[e], 0.f);
                }
            }
        T *o = reinterpret_cast<T *>(p.out) + n * p.out_img_stride + ((long long)oy * p.Wo + ox) * p.out_ld + cc * STEP;
        if (VEC) {
            T tv[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) tv[e] = (T)best[e];
            uint4v u;
            __builtin_memcpy(&u, tv, 16);
            *reinterpret_cast<uint4v *>(o) = u;
        } else {
            o[0] = (T)best[0];
        }
    }
}

// Generic fallback, one element per thread: out[map(n,y,x)][c] = a[n,y,x,c] (+ b[n,y,x,c]).
// map: identity, nearest upsample x2 (layers.py:112-116) or block-major reorg x2 (layers.py:90-97).
template <bool F32>
__global__ void __launch_bounds__(256) eltwise_kernel(const EltParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < p.total; w += stride) {
        const int c = (int)(w % p.C);
        long long t = w / p.C;
        const int x = (int)(t % p.W); t /= p.W;
        const int y = (int)(t % p.H);
        const long long n = t / p.H;
        const long long pa = n * p.a_img_stride + ((long long)y * p.W + x) * p.a_ld + c;
        float v = p.a_f32 ? reinterpret_cast<const float *>(p.a)[pa] : (float)reinterpret_cast<const T *>(p.a)[pa];
        if (p.b) {
            const float r = (float)reinterpret_cast<const T *>(p.b)[n * p.b_img_stride + ((long long)y * p.W + x) * p.b_ld + c];
            // the reference adds two tensors of type T: round the sum once in T
            v = v + r;
        }
        long long off[4];
        int npos = 1;
        if (p.outmode == OUT_NORMAL) {
            off[0] = n * p.out_img_stride + ((long long)y * p.W + x) * p.out_ld + c;
        } else if (p.outmode == OUT_UP2) {
            const long long W2 = 2LL * p.W;
            const long long b0 = n * p.out_img_stride + ((2LL * y) * W2 + 2LL * x) * p.out_ld + c;
            off[0] = b0; off[1] = b0 + p.out_ld; off[2] = b0 + W2 * p.out_ld; off[3] = b0 + (W2 + 1) * p.out_ld;
            npos = 4;
        } else {
            const int W2 = p.W >> 1;
            off[0] = n * p.out_img_stride + ((long long)(y >> 1) * W2 + (x >> 1)) * p.out_ld + ((y & 1) * 2 + (x & 1)) * p.C + c;
        }
        for (int q = 0; q < npos; ++q) {
            if (p.out_f32) reinterpret_cast<float *>(p.out)[off[q]] = v;
            else reinterpret_cast<T *>(p.out)[off[q]] = (T)v;
        }
    }
}

static inline unsigned grid_for(long long work) {
    long long g = (work + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;     // 16 blocks per CU, grid-stride the rest
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Image preprocessing of the TEST loop (reference net/base.py:115-155: cv2.resize INTER_LINEAR stretch, BGR->RGB, / 255.):
// one thread per destination pixel.  8-bit INTER_LINEAR as OpenCV defines it (imgproc/resize.cpp): half-pixel centres,
// weights rounded to 11-bit fixed point, horizontal pass in int32, vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; then float32(v / 255.) (division in float64 like NumPy).
__device__ __forceinline__ void resize_coeff(int d, int src, int dst, int &s0, int &s1, int &w0, int &w1) {
#pragma clang fp contract(off)
    const double scale = (double)src / (double)dst;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    w0 = __float2int_rn((1.f - f) * 2048.f);        // cvRound: round half to even
    w1 = __float2int_rn(f * 2048.f);
    s0 = s;
    s1 = s + 1 < src ? s + 1 : src - 1;
}

__global__ void __launch_bounds__(256) resize_u8_kernel(const ResizeParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.dst_h * p.dst_w) return;
    const int dy = (int)(i / p.dst_w), dx = (int)(i - (long long)dy * p.dst_w);
    int x0, x1, a0, a1, y0, y1, b0, b1;
    resize_coeff(dx, p.src_w, p.dst_w, x0, x1, a0, a1);
    resize_coeff(dy, p.src_h, p.dst_h, y0, y1, b0, b1);
    const unsigned char *r0 = p.src + (long long)y0 * p.src_row_bytes, *r1 = p.src + (long long)y1 * p.src_row_bytes;
    const bool same = p.src_h == p.dst_h && p.src_w == p.dst_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = p.swap_rb ? 2 - c : c;
        int v;
        if (same) {
            v = r0[x0 * 3 + cs];
        } else {
            const int h0 = (int)r0[x0 * 3 + cs] * a0 + (int)r0[x1 * 3 + cs] * a1;
            const int h1 = (int)r1[x0 * 3 + cs] * a0 + (int)r1[x1 * 3 + cs] * a1;
            v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        p.dst[i * 3 + c] = (float)((double)v / 255.);
    }
}

hipError_t launch_resize(const ResizeParams &p, hipStream_t s) {
    const long long n = (long long)p.dst_h * p.dst_w;
    if (n <= 0 || p.src_h <= 0 || p.src_w <= 0 || (n + 255) / 256 > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_prep(const PrepParams &p, int dtype, hipStream_t s) {
    if (dtype == YOLO_DTYPE_F16) hipLaunchKernelGGL(prep_kernel<false>, dim3(grid_for(p.pixels)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(prep_kernel<true>, dim3(grid_for(p.pixels)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_pool(const PoolParams &p0, int dtype, hipStream_t s) {
    PoolParams p = p0;
    const int epc = dtype == YOLO_DTYPE_F16 ? 8 : 4;
    const bool vec = (p.C % epc == 0) && (p.in_ld % epc == 0) && (p.out_ld % epc == 0) &&
                     ((uintptr_t)p.in % 16 == 0) && ((uintptr_t)p.out % 16 == 0) &&
                     (p.in_img_stride % epc == 0) && (p.out_img_stride % epc == 0);
    const long long pix = p.total;      // caller passes B*Ho*Wo
    p.total = pix * (vec ? p.C / epc : p.C);
    const dim3 g(grid_for(p.total)), b(256);
    if (dtype == YOLO_DTYPE_F16) {
        if (vec) hipLaunchKernelGGL((pool_kernel<false, true>), g, b, 0, s, p);
        else hipLaunchKernelGGL((pool_kernel<false, false>), g, b, 0, s, p);
    } else {
        if (vec) hipLaunchKernelGGL((pool_kernel<true, true>), g, b, 0, s, p);
        else hipLaunchKernelGGL((pool_kernel<true, false>), g, b, 0, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_eltwise(const EltParams &p, int dtype, hipStream_t s) {
    const dim3 g(grid_for(p.total)), b(256);
    if (dtype == YOLO_DTYPE_F16) hipLaunchKernelGGL(eltwise_kernel<false>, g, b, 0, s, p);
    else hipLaunchKernelGGL(eltwise_kernel<true>, g, b, 0, s, p);
    return hipGetLastError();
}

// Split-K second pass: out = epilogue(sum over the K splits of the raw float32 accumulators), with the epilogue semantics of
// conv_common.h: + folded-BN bias -> leaky 0.1 -> + residual (no activation after the add) -> output index map (identity /
// nearest-upsample x2 / block-major reorg) -> T or float32; head convs also fill the compact objectness array.  One thread per
// (pixel, 4 couts); the slabs are a few MB and L2-resident.
template <bool F32>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ReduceParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    typedef float float4v __attribute__((ext_vector_type(4)));
    const int groups = p.cout_pad >> 2;
    const long long total = (long long)p.M * groups;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int m = (int)(idx / groups);
        const int c = (int)(idx - (long long)m * groups) * 4;
        if (c >= p.Cout) continue;
        const float *src = p.part + (size_t)m * p.cout_pad + c;
        float4v a = *reinterpret_cast<const float4v *>(src);
        for (int s = 1; s < p.ksplit; ++s) a += *reinterpret_cast<const float4v *>(src + (size_t)s * p.M * p.cout_pad);
        const int n = m / p.HoWo, rem = m - n * p.HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        float v[4] = {a.x, a.y, a.z, a.w};
        const int nv = p.Cout - c < 4 ? p.Cout - c : 4;
        for (int i = 0; i < nv; ++i) {
            const float x = v[i] + p.bias[c + i];
            v[i] = p.leaky ? fmaxf(0.1f * x, x) : x;
        }
        if (p.res) {
            const T *rp = reinterpret_cast<const T *>(p.res) + (long long)n * p.res_img_stride + (long long)rem * p.res_ld + c;
            for (int i = 0; i < nv; ++i) v[i] += (float)rp[i];
        }
        long long off[4];
        int npos = 1;
        if (p.outmode == OUT_NORMAL) {
            off[0] = (long long)n * p.out_img_stride + (long long)rem * p.out_ld + c;
        } else if (p.outmode == OUT_UP2) {
            const long long W2 = 2LL * p.Wo;
            const long long base = (long long)n * p.out_img_stride + ((2LL * oy) * W2 + 2LL * ox) * p.out_ld + c;
            off[0] = base; off[1] = base + p.out_ld; off[2] = base + W2 * p.out_ld; off[3] = base + (W2 + 1) * p.out_ld;
            npos = 4;
        } else {
            const int W2 = p.Wo >> 1;
            off[0] = (long long)n * p.out_img_stride + ((long long)(oy >> 1) * W2 + (ox >> 1)) * p.out_ld + ((oy & 1) * 2 + (ox & 1)) * p.Cout + c;
        }
        for (int q = 0; q < npos; ++q) {
            if (p.out_f32) {
                float *op = reinterpret_cast<float *>(p.out) + off[q];
                for (int i = 0; i < nv; ++i) op[i] = v[i];
            } else {
                T *op = reinterpret_cast<T *>(p.out) + off[q];
                for (int i = 0; i < nv; ++i) op[i] = (T)v[i];
            }
        }
        if (p.obj_out) {
            for (int i = 0; i < nv; ++i) {
                const int a_ = (c + i) / p.obj_width;
                if (c + i - a_ * p.obj_width == 4) p.obj_out[n * p.obj_rows + p.obj_row0 + rem * p.obj_na + a_] = v[i];
            }
        }
    }
}

hipError_t launch_splitk_reduce(const ReduceParams &p, hipStream_t s) {
    if (p.ksplit < 2 || !p.part || (p.cout_pad & 3) || p.M <= 0) return hipErrorInvalidValue;
    const dim3 g(grid_for((long long)p.M * (p.cout_pad >> 2))), b(256);
    if (p.f32) hipLaunchKernelGGL(splitk_reduce_kernel<true>, g, b, 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, g, b, 0, s, p);
    return hipGetLastError();
}

// the names rocprofv3's kernel trace prints (yolo_kernel_info.symbol)
std::string aux_symbol(int kind, int dtype, bool vec) {
    const char *f = dtype == YOLO_DTYPE_F16 ? "false" : "true";
    if (kind == K_PREP) return std::string("void yolo::prep_kernel<") + f + ">(yolo::PrepParams)";
    if (kind == K_POOL) return std::string("void yolo::pool_kernel<") + f + ", " + (vec ? "true" : "false") + ">(yolo::PoolParams)";
    return std::string("void yolo::eltwise_kernel<") + f + ">(yolo::EltParams)";
}

}  // namespace yolo
