// C ABI of libyolo_hip.so (declared in include/yolo_hip.h).  No exceptions cross the boundary.
#include <cstdio>
#include <cstring>
#include <new>

#include <cmath>
#include "yolo_internal.h"

namespace yolo {
const char *get_error();
}
using namespace yolo;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
            return YOLO_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

static int fail(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

namespace {
// images one part of a full batch holds (what every launch of a forward pass sees at most)
inline int part_batch(const yolo_net *net) { return (net->opt.max_batch + net->parts - 1) / net->parts; }
const size_t kPairCounterBytes = 65536;            // in-launch pair / split-K (conv_tap.hip): one ticket per tile, a 128-byte line each (512 tiles), in front of the slabs

// packed weights of a kernel; nullptr while no weights are bound (yolo_net_kernel_info / describe build launch parameters of a plan that
// has none yet: no offset is applied to a null pointer -- found by the sanitizer build, tests/test_sanitizer.py)
static inline const unsigned char *weights_at(const yolo_net *net, size_t off) { return net->dev_weights ? net->dev_weights + off : nullptr; }
bool conv_tile_valid(const yolo_net *net, const Kernel &k, int tile);
size_t splitk_slab_bytes(const yolo_net *net);

// Ticket counters of the in-launch pair split: every launch returns them to zero, so they are cleared when the workspace is bound
// and again after any failed forward (a launch that did not run may leave the forward half-way).  The memset goes to the null
// stream, which does not order against the non-blocking streams a caller may launch on: hence the device-wide synchronise.
int zero_pair_counters(yolo_net *net) {
    if (!net->splitk_bytes || !net->dev_ws) return YOLO_OK;
    const size_t per = net->splitk_bytes / (size_t)net->arenas;
    for (int a = 0; a < net->arenas; ++a)
        HIP_TRY(hipMemset(net->dev_ws + net->splitk_off + (size_t)a * (per / 256 * 256), 0, per < kPairCounterBytes ? per : kPairCounterBytes));
    HIP_TRY(hipDeviceSynchronize());
    return YOLO_OK;
}
}

extern "C" {

int yolo_hip_abi_version(void) { return YOLO_HIP_ABI_VERSION; }
const char *yolo_last_error(void) { return get_error(); }

int yolo_net_create(const yolo_layer_desc *layers, int n_layers, const yolo_net_options *opt, yolo_net **out) {
    if (!layers || !opt || !out || n_layers <= 0) return fail(YOLO_ERR_ARG, "yolo_net_create: null argument");
    if (opt->dtype != YOLO_DTYPE_F16 && opt->dtype != YOLO_DTYPE_F32) return fail(YOLO_ERR_ARG, "yolo_net_create: bad dtype");
    if (opt->max_batch <= 0) return fail(YOLO_ERR_ARG, "yolo_net_create: max_batch must be positive");
    yolo_net *net = new (std::nothrow) yolo_net();
    if (!net) return fail(YOLO_ERR_ARG, "out of host memory");
    net->opt = *opt;
    if (net->opt.cand_capacity <= 0) net->opt.cand_capacity = 4096;
    if (net->opt.max_boxes <= 0) net->opt.max_boxes = 256;
    if (net->opt.cand_capacity > 65536) {
        delete net;
        return fail(YOLO_ERR_ARG, "cand_capacity above 65536 (16-bit sort indices)");
    }
    std::string err;
    int rc = YOLO_ERR_PLAN;
    try {
        rc = plan_network(net, layers, n_layers, err);
    } catch (const std::exception &e) {
        err = e.what();
    }
    if (rc != YOLO_OK) {
        delete net;
        return fail(rc, "yolo_net_create: " + err);
    }
    if (net->opt.force_tile > 0)        // test / tuning hook: one tile id on every conv that accepts it
        for (Kernel &k : net->kernels)
            if (k.kind == K_CONV && k.stem < 2 && conv_tile_valid(net, k, net->opt.force_tile - 1)) k.tile = net->opt.force_tile - 1;
    // the split-K slab is the last region of the workspace (plan.cpp reserves nothing for it): sized from the launches that can split
    net->splitk_bytes = splitk_slab_bytes(net) * (size_t)net->arenas;
    net->workspace_bytes = net->splitk_off + net->splitk_bytes;
    *out = net;
    return YOLO_OK;
}

void yolo_net_destroy(yolo_net *net) {
    if (!net) return;
    if (net->e_fork) (void)hipEventDestroy(net->e_fork);
    for (hipEvent_t e : net->e_join) (void)hipEventDestroy(e);
    for (hipStream_t st : net->side) (void)hipStreamDestroy(st);
    for (hipEvent_t e : net->e_bfork) (void)hipEventDestroy(e);
    for (hipEvent_t e : net->e_bjoin) (void)hipEventDestroy(e);
    for (hipStream_t st : net->branch) (void)hipStreamDestroy(st);
    delete net;
}

size_t yolo_net_weight_count(const yolo_net *net) { return net ? net->weight_count : 0; }
size_t yolo_net_weights_bytes(const yolo_net *net) { return net ? net->weights_bytes : 0; }
size_t yolo_net_workspace_bytes(const yolo_net *net) { return net ? net->workspace_bytes : 0; }
size_t yolo_net_output_count(const yolo_net *net) { return net ? net->out_count : 0; }
double yolo_net_flops_per_image(const yolo_net *net) { return net ? net->flops_per_image : 0.0; }
int yolo_net_num_kernels(const yolo_net *net) { return net ? (int)net->kernels.size() : 0; }
int yolo_net_num_streams(const yolo_net *net) { return net ? net->parts : 0; }

int yolo_net_head_desc(const yolo_net *net, yolo_head_desc *out) {
    if (!net || !out) return fail(YOLO_ERR_ARG, "yolo_net_head_desc: null argument");
    *out = net->head;
    return YOLO_OK;
}

static int check_head(const yolo_head_desc *h, size_t out_count, std::string &err) {
    if (h->version != 2 && h->version != 3) { err = "head version must be 2 or 3"; return YOLO_ERR_ARG; }
    if (h->n_scales < 1 || h->n_scales > YOLO_MAX_SCALES || h->n_classes < 1) { err = "bad head geometry"; return YOLO_ERR_ARG; }
    size_t rows = 0;
    for (int s = 0; s < h->n_scales; ++s) {
        if (h->n_anchors[s] < 1 || h->n_anchors[s] > YOLO_MAX_ANCHORS || h->h[s] < 1 || h->w[s] < 1) { err = "bad head scale"; return YOLO_ERR_ARG; }
        rows += (size_t)h->h[s] * h->w[s] * h->n_anchors[s];
    }
    if (out_count && rows * (5 + h->n_classes) != out_count) { err = "head geometry does not match the network output size"; return YOLO_ERR_ARG; }
    return YOLO_OK;
}

int yolo_net_set_head(yolo_net *net, const yolo_head_desc *head) {
    if (!net || !head) return fail(YOLO_ERR_ARG, "yolo_net_set_head: null argument");
    std::string err;
    int rc = check_head(head, net->out_count, err);
    if (rc) return fail(rc, "yolo_net_set_head: " + err);
    net->head = *head;
    return YOLO_OK;
}

int yolo_net_workspace_regions(const yolo_net *net, yolo_ws_region *out, int cap) {
    if (!net) return 0;
    int n = 0;
    auto add = [&](const std::string &name, size_t off, size_t used, size_t region) {
        if (out && n < cap) {
            memset(&out[n], 0, sizeof out[n]);
            snprintf(out[n].name, sizeof out[n].name, "%s", name.c_str());
            out[n].offset = off; out[n].used_bytes = used; out[n].region_bytes = region;
        }
        ++n;
    };
    for (int a = 0; a < net->arenas; ++a)
        for (size_t b = 0; b < net->buffers.size(); ++b) {
            const Buffer &B = net->buffers[b];
            if (!B.bytes) continue;
            add("tensor " + std::to_string(b) + " arena " + std::to_string(a), (size_t)a * net->arena_bytes + B.offset, B.used, B.bytes);
        }
    const size_t mb = (size_t)net->opt.max_batch;
    add("head logits", net->logits_off, net->out_count * 4 * mb, net->cand_off - net->logits_off);
    add("candidates", net->cand_off, sizeof(Candidate) * (size_t)net->opt.cand_capacity * mb, net->count_off - net->cand_off);
    add("candidate counters", net->count_off, sizeof(int) * mb * kCandCountStride, net->nms_off - net->count_off);
    add("nms scratch", net->nms_off, nms_scratch_bytes(net->opt.cand_capacity) * mb, net->obj_off - net->nms_off);
    add("objectness", net->obj_off, net->obj_bytes, net->splitk_off - net->obj_off);
    if (net->splitk_bytes) add("split-K tickets + slabs", net->splitk_off, net->splitk_bytes, net->splitk_bytes);
    return n;
}

size_t yolo_net_describe(const yolo_net *net, char *buf, size_t cap) {
    if (!net) return 0;
    std::string s = describe(net);
    if (buf && cap) {
        size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size() + 1;
}

int yolo_net_load_weights(yolo_net *net, const float *host_weights, size_t n, void *dev_weights, size_t dev_bytes) {
    if (!net || !dev_weights || (!host_weights && n)) return fail(YOLO_ERR_ARG, "yolo_net_load_weights: null argument");
    if (dev_bytes < net->weights_bytes) return fail(YOLO_ERR_ARG, "yolo_net_load_weights: device buffer too small");
    if ((uintptr_t)dev_weights % 256) return fail(YOLO_ERR_ARG, "yolo_net_load_weights: device buffer must be 256-byte aligned");
    std::vector<unsigned char> blob;
    std::string err;
    int rc = pack_weights(net, host_weights, n, blob, err);
    if (rc) return fail(rc, "yolo_net_load_weights: " + err);
    HIP_TRY(hipMemcpy(dev_weights, blob.data(), blob.size(), hipMemcpyHostToDevice));
    net->dev_weights = static_cast<unsigned char *>(dev_weights);
    net->weights_loaded = true;
    return YOLO_OK;
}

int yolo_net_bind_workspace(yolo_net *net, void *ws, size_t bytes) {
    if (!net || !ws) return fail(YOLO_ERR_ARG, "yolo_net_bind_workspace: null argument");
    if (bytes < net->workspace_bytes) return fail(YOLO_ERR_ARG, "yolo_net_bind_workspace: workspace too small");
    if ((uintptr_t)ws % 256) return fail(YOLO_ERR_ARG, "yolo_net_bind_workspace: workspace must be 256-byte aligned");
    net->dev_ws = static_cast<unsigned char *>(ws);
    net->cand_clean = 0;        // (whatever the new workspace holds where the candidate counters live)
    net->dev_ws_bytes = bytes;
    int rc = zero_pair_counters(net);
    if (rc) return rc;
    return YOLO_OK;
}

}  // extern "C"

// ---- forward ------------------------------------------------------------------------------------
namespace {

struct Ptrs {
    yolo_net *net;
    const float *in;            // already advanced to image `img0` by the caller
    float *out;                 // ditto
    int img0 = 0;               // first image of this pass in the batch (two-stream halves: compact per-image arrays)
    int arena = 0;              // activation arena of this pass
    unsigned char *buf_base(int b) const {
        if (b == BUF_USER_IN) return reinterpret_cast<unsigned char *>(const_cast<float *>(in));
        if (b == BUF_USER_OUT) return reinterpret_cast<unsigned char *>(out);
        return net->dev_ws + (size_t)arena * net->arena_bytes + net->buffers[b].offset;
    }
    int esz(const View &v) const { return v.f32 ? 4 : net->esize; }
    // element pointer of channel 0 of pixel 0 of image 0 of the view
    unsigned char *view_ptr(const View &v) const { return buf_base(v.buf) + (size_t)(v.base + v.coff) * esz(v); }
};

// 16-byte epilogue accesses are possible when every stride of the view is chunk-aligned (planned buffers start 4096-byte aligned
// inside a 256-byte aligned workspace; a caller-owned tensor is checked at launch)
void conv_vec_flags(const yolo_net *net, const Kernel &k, bool out_f32, int &vec_out, int &vec_res) {
    const int epc = net->epc;
    const int ch = k.cfg == CFG_N32 ? 8 : 16;
    const int oepc = out_f32 ? 4 : epc;
    vec_out = (k.cout % ch == 0) && (k.out.ld % oepc == 0) && ((k.out.base + k.out.coff) % oepc == 0) && (k.out.img_stride % oepc == 0);
    vec_res = k.has_res && (k.cout % ch == 0) && (k.in2.ld % epc == 0) && (k.in2.coff % epc == 0) && (k.in2.img_stride % epc == 0);
}

// conv launch parameters for one planned kernel at the given batch
int make_conv_params(yolo_net *net, const Kernel &k, const Ptrs &P, int batch, ConvParams &p) {
    const int dtype = net->opt.dtype;
    memset(&p, 0, sizeof p);
    const View &in = k.in;
    const long long in_bytes = (long long)batch * in.img_stride * net->esize;
    if (in_bytes > 0x7ffffff0LL)
        return fail(YOLO_ERR_ARG, "conv input tensor exceeds 2 GiB (32-bit buffer addressing): lower the batch");
    p.in = P.buf_base(in.buf);
    p.in_bytes = (uint32_t)in_bytes;
    p.wgt = weights_at(net, k.w_off);
    p.wgt_bytes = (uint32_t)k.w_bytes;
    p.bias = reinterpret_cast<const float *>(weights_at(net, k.b_off));
    p.H = in.H; p.W = in.W; p.in_ld = in.ld; p.in_coff = in.coff; p.in_img_stride = in.img_stride;
    const yolo_layer_desc &d = net->layers[k.src_layer].d;
    const int Ho = net->layers[k.src_layer].H, Wo = net->layers[k.src_layer].W;
    p.Ho = Ho; p.Wo = Wo; p.HoWo = Ho * Wo;
    const long long M = (long long)batch * Ho * Wo;
    if (M > 0x7fffffffLL) return fail(YOLO_ERR_ARG, "too many output pixels for one launch");
    p.M = (int)M;
    p.Cout = k.cout;
    p.out = P.view_ptr(k.out);
    p.out_ld = k.out.ld;
    p.out_img_stride = k.out.img_stride;
    p.out_f32 = k.out.f32 || dtype == YOLO_DTYPE_F32;
    {   // extents for buffer-addressed epilogues (conv_tap.hip stream kernel): 0 when a tensor is not below 2 GiB
        const long long ob = (long long)batch * k.out.img_stride * (p.out_f32 ? 4 : net->esize);
        p.out_bytes = ob > 0 && ob <= 0x7ffffff0LL ? (uint32_t)ob : 0u;
    }
    p.ksize = d.ksize; p.stride = d.stride; p.pad = (d.ksize - 1) / 2; p.taps = d.ksize * d.ksize;
    p.ktiles = k.ktiles;
    p.tiles_per_tap = k.perchunk ? 1 : k.cpt / 8;
    p.cin_chunks = k.cpt;
    p.cpt_shift = k.cpt == 1 ? 0 : k.cpt == 2 ? 1 : 2;
    p.wrow_bytes = (uint32_t)k.ktiles * 128;
    p.leaky = k.leaky; p.outmode = k.outmode; p.has_res = k.has_res;
    p.f32 = dtype == YOLO_DTYPE_F32;
    int vo = 0, vr = 0;
    conv_vec_flags(net, k, p.out_f32 != 0, vo, vr);
    p.vec_out = vo && ((uintptr_t)P.buf_base(k.out.buf) % 16 == 0);
    if (k.head && net->obj_bytes && net->head.n_classes > 0 && !p.vec_out && p.out_f32 && k.outmode == OUT_NORMAL && !k.has_res) {
        const int width = 5 + net->head.n_classes;
        const long long base = k.out.base + k.out.coff;
        const size_t rows = net->out_count / (size_t)width;
        if (k.out.ld % width == 0 && base % width == 0 && rows * width == net->out_count && (size_t)batch * rows * 4 <= net->obj_bytes) {
            p.obj_out = reinterpret_cast<float *>(net->dev_ws + net->obj_off) + (size_t)P.img0 * rows;
            p.obj_width = width; p.obj_rows = (int)rows; p.obj_row0 = (int)(base / width); p.obj_na = k.out.ld / width;
            p.obj_min = net->obj_min_logit;     // -inf outside yolo_net_detect: every row is written
        }
    }
    if (k.has_res) {
        p.res = P.view_ptr(k.in2);
        p.res_ld = k.in2.ld;
        p.res_img_stride = k.in2.img_stride;
        const long long rbytes = (long long)batch * k.in2.img_stride * net->esize;
        p.res_bytes = rbytes > 0 && rbytes <= 0x7ffffff0LL ? (uint32_t)rbytes : 0u;
        p.vec_res = vr;
    }
    return YOLO_OK;
}

// the LDS-DMA kernels take fp16 convs whose Cin is a multiple of 4 chunks (32 channels)
// float32 nets: only the tap-reuse kernel (3x3/1) has a float32 instantiation besides the 4-wave kernel
bool dma_eligible(const yolo_net *net, const Kernel &k) {
    if (k.cpt % 4) return false;
    return net->opt.dtype == YOLO_DTYPE_F16 || (k.ksize == 3 && k.stride == 1);
}
// tile 0 = the 4-wave kernel of conv.hip with the planner's cfg (always available)
bool conv_tile_valid(const yolo_net *net, const Kernel &k, int tile) {
    if (k.outmode == OUT_POOL2 && tile != 12 && tile != 13 && tile != 17) return false;      // the fused max-pool lives in the 16 x 16 2-D tap tiles
    if (tile == 0) return true;
    if (net->opt.dtype == YOLO_DTYPE_F32 && !dma_cfg_f32_ok(tile)) return false;
    if ((tile == 18 || tile == 21 || tile == 22) && k.in.H != k.in.W) return false;     // the image-aligned tap tiles: square maps (the rules price tiles by W alone)
    if ((tile == 20 || tile == 21 || tile == 23) && ((k.in.H & 1) || net->opt.dtype != YOLO_DTYPE_F16)) return false;      // stride 2 over parity planes: even maps, fp16
    return dma_eligible(net, k) && dma_cfg_valid(tile, k.cout, k.cpt, true, k.ksize, k.stride, k.in.W);
}

// Split-K decision for one conv launch (0 = the 4-wave kernel with the planner's cfg, else a tap-reuse tile): when the launch
// would leave most of the chip idle (<= 128 workgroups) and K is long, the K range is cut into `ks` splits of `ku` units
// (conv_tap.hip: channel slices, conv.hip: K tiles) so that ~384 workgroups exist; their float32 partial sums meet in
// splitk_reduce_kernel.  Returns 1 when the launch stays whole.
int choose_ksplit(const Kernel &k, const ConvParams &p, int tile, size_t slab_bytes, int &ku) {
    ku = 0;
    if (!slab_bytes || p.M <= 0) return 1;
    long long blocks;
    int units, min_units;
    if (tile == 0) {
        const int na = k.cfg == CFG_N128 ? 128 : k.cfg == CFG_N64 ? 64 : 32, nb = k.cfg == CFG_N128 ? 128 : 256;
        blocks = ((long long)p.M + nb - 1) / nb * ((p.Cout + na - 1) / na);
        units = p.ktiles;
        min_units = 2;                  // >= 64 (float32) / 128 (fp16) k per split: these launches are latency-bound, not MFMA-bound
    } else if (dma_cfg_is_tap(tile) && dma_cfg_splitk_ok(tile)) {
        const long long mq = (long long)(p.M / p.HoWo) * (p.H + 1) * (p.W + 1);        // padded-linear positions
        blocks = (mq + dma_cfg_nb(tile) - 1) / dma_cfg_nb(tile) * ((p.Cout + dma_cfg_na(tile) - 1) / dma_cfg_na(tile));
        units = p.cin_chunks >> 2;
        min_units = 2;                  // >= 288 (float32) / 576 (fp16) k per split
    } else {
        return 1;
    }
    if (blocks > 128 || units < 2 * min_units) return 1;
    // workgroups = blocks x ks: 512 (two per CU) when K is long enough for that many splits, else 256, else whatever K allows --
    // a count between the two leaves some CUs with two workgroups and the rest with one, and the pairs set the time
    const long long kmax = units / min_units < 32 ? units / min_units : 32;
    long long ks = 512 / blocks;
    if (ks > kmax) ks = 256 / blocks;
    if (ks > kmax) ks = kmax;
    const size_t cout_pad = ((size_t)p.Cout + 127) / 128 * 128;
    // the partial sums are written and read back once: worth it while that traffic stays in the order of the weight stream the
    // launch reads anyway (measured: YOLOv2 13x13 at batch 1, 22 MB of partials beside 38 MB of weights, 553 -> 70 us; YOLOv3 19x19
    // at batch 8, 47 MB beside 9 MB, slower than unsplit); a few MB are always fine (L2-resident, ~2 us)
    const size_t wbytes = (size_t)p.Cout * (size_t)p.taps * (size_t)p.cin_chunks * 16;
    // (1x1 layers: up to 24 MB -- tiny-YOLOv2's head 1024 -> 125 at 13x13, batch 64, is 85 workgroups walking K = 1024 alone: 85 us whole,
    // 53 us as four splits with 22 MB of partial sums)
    // 3x3: 8 MB; 16 MB where a split still walks a long K loop -- float32 (MFMA 16x slower: YOLOv2-416 b1 104 x 104 64 -> 128 43 -> 33 us,
    // 52 x 52 and 26 x 26 layers 40 -> 33 us, step 0.815 -> 0.757 ms) or >= 8 channel slices (YOLOv3-608 b1 38 x 38: 2 -> 4 splits, 19.6 -> 18 us);
    // a short-K fp16 layer loses with it (76 x 76 128 -> 256 at batch 1: 15.7 -> 18.8 us)
    const size_t small_cap = (size_t)(p.taps == 1 ? 24 : (p.f32 || units >= 8) ? 16 : 8) << 20;
    while (ks >= 2 && ((size_t)ks * (size_t)p.M * cout_pad * 4 > slab_bytes || (size_t)ks * (size_t)p.M * cout_pad * 4 > (2 * wbytes > small_cap ? 2 * wbytes : small_cap))) --ks;
    if (ks < 2) return 1;
    ku = (int)((units + ks - 1) / ks);
    return (units + ku - 1) / ku;       // every split owns at least one unit
}

// What one conv launch runs: the tile (0 = the 4-wave kernel of conv.hip with the planner's cfg, > 0 = conv_dma.hip tile id) and
// the K split (ks = 1: whole K).  tile_req < 0: the rules of choose_dma_cfg.  One function for the launch path, the workspace
// sizing (split-K slab) and yolo_net_kernel_info, so what is reported is what runs.
struct ConvPick { int tile, ks, ku, pair; };
const size_t kSplitkSlabMax = (size_t)64 << 20;     // per arena

ConvPick pick_conv(const yolo_net *net, const Kernel &k, const ConvParams &p, int tile_req, size_t slab_bytes) {
    int tile = tile_req;
    if (!dma_eligible(net, k) || (tile > 0 && !conv_tile_valid(net, k, tile))) tile = 0;
    else if (tile < 0) {
        tile = choose_dma_cfg(p.M, k.cout, k.cpt, p.taps, k.has_res, true, k.stride, k.in.W, net->opt.dtype == YOLO_DTYPE_F32);
        if (tile == 18 && !conv_tile_valid(net, k, 18)) tile = conv_tile_valid(net, k, 15) ? 15 : 8;
        if (tile == 22 && !conv_tile_valid(net, k, 22)) tile = conv_tile_valid(net, k, 10) ? 10 : 8;
        if ((tile == 20 || tile == 21) && !conv_tile_valid(net, k, tile)) tile = conv_tile_valid(net, k, 20) ? 20 : conv_tile_valid(net, k, 5) ? 5 : 0;
        if (tile == 23 && !conv_tile_valid(net, k, 23)) tile = conv_tile_valid(net, k, 6) ? 6 : 0;
    }
    // A launch for the in-launch pair on a WIDE tile (64-128 tiles of 128 x 256, or of the image-aligned 128 x 192: half the weight
    // bytes per flop of the 128 x 128 tile) is not split any other way.
    static const bool no_pair_w = getenv("YOLO_NO_PAIR_SPLIT") != nullptr;
    bool wide_pair = false;
    if (tile > 0 && dma_cfg_is_tap(tile) && !no_pair_w && tile_req <= 0 && !p.f32 && (p.cin_chunks >> 2) >= 8 && p.HoWo > 0) {
        const long long mq = (long long)(p.M / p.HoWo) * (p.H + 1) * (p.W + 1);
        const long long ct = (p.Cout + 127) / 128;
        const long long b8 = (mq + 255) / 256 * ct, b22 = (long long)(p.M / p.HoWo) * ct;
        wide_pair = (conv_tile_valid(net, k, 22) && b22 >= 64 && b22 <= 128 && (size_t)b22 * 2 * 98304 <= slab_bytes) ||
                    (conv_tile_valid(net, k, 8) && b8 >= 64 && b8 <= 128 && (size_t)b8 * 2 * 131072 <= slab_bytes);
    }
    int ku = 0;
    int ks = wide_pair ? 1 : choose_ksplit(k, p, tile, slab_bytes, ku);
    // a 3x3/1 layer small enough for split-K runs it on the 128 x 128 tap tile (the one with the split-K instantiation), whatever
    // tile the cost model would pick for the whole-K launch
    // (an explicitly requested tile -- force_tile, an autotune candidate -- runs as requested)
    if (ks <= 1 && !wide_pair && tile_req <= 0 && tile > 0 && dma_cfg_is_tap(tile) && tile != 11 && conv_tile_valid(net, k, 11)) {
        int ku11 = 0;
        const int ks11 = choose_ksplit(k, p, 11, slab_bytes, ku11);
        if (ks11 > 1) { tile = 11; ks = ks11; ku = ku11; }
    }
    // 129-256 tiles of 128 x 128 (13 x 13 / 19 x 19 maps at batch 8-32) with a long K: every workgroup would run ALONE on its CU at
    // 0.6 of the rate a pair reaches (block trace, profiles/r03_ablation.md).  K in two halves inside ONE launch (conv_tap.hip):
    // two co-resident half-K workgroups per tile, the second arriver sums -- no reduce kernel, 2 x 64 KiB of slab per tile.
    int pair = 0;
    static const bool no_pair = getenv("YOLO_NO_PAIR_SPLIT") != nullptr;       // A/B switch (read once; results unchanged up to summation order)
    // (an explicitly requested tile -- force_tile, an autotune candidate -- runs as requested: the hook must time and test the tile it names)
    if (ks <= 1 && tile > 0 && dma_cfg_is_tap(tile) && !no_pair && tile_req <= 0) {
        const long long mq = (long long)(p.M / p.HoWo) * (p.H + 1) * (p.W + 1);
        const int units = p.cin_chunks >> 2;
        const long long ct = (p.Cout + 127) / 128;
        // the 128 x 256 tile first (fp16): per CU the K loop of the 128 x 128 tile is bound by the LDS-DMA path (8 KiB of weights
        // per tap for 1 MFLOP: two half-K workgroups on a CU were measured no faster than one whole-K one), the wider tile halves
        // the weight bytes per flop
        const long long b8 = (mq + 255) / 256 * ct, b11 = (mq + 127) / 128 * ct;
        const long long b22 = (long long)(p.M / p.HoWo) * ct;        // image-aligned 128 x 192 tile: a tile per image and cout tile
        // (that instantiation is built for ONE workgroup per CU -- it needs 180 registers --, so at most 128 tiles = 256 half-K workgroups:
        // 26 x 26 at batch 16 = 184 tiles ran 48 us as 368 halves against 33 us whole)
        // (12 x 12 / 13 x 13 maps: one image per 192-position tile -- 5 % padding where 256-position tiles of the padded-linear grid
        // compute 23 %, and 16 images x 8 cout tiles x 2 halves are exactly 256 workgroups: YOLOv2-416 b16 13 x 13 layers -25 %)
        if (units >= 8 && !p.f32 && conv_tile_valid(net, k, 22) && b22 >= 64 && b22 <= 128 && (size_t)b22 * 2 * 98304 <= slab_bytes) {
            tile = 22; ks = 2; ku = (units + 1) / 2; pair = 1;
        } else if (units >= 8 && !p.f32 && conv_tile_valid(net, k, 8) && b8 >= 64 && b8 <= 128 && (size_t)b8 * 2 * 131072 <= slab_bytes) {
            tile = 8; ks = 2; ku = (units + 1) / 2; pair = 1;
        } else if (units >= 8 && conv_tile_valid(net, k, 11) && b11 > 128 && b11 <= 256 && (size_t)b11 * 2 * 65536 <= slab_bytes) {
            tile = 11; ks = 2; ku = (units + 1) / 2; pair = 1;
        }
    }
    // Split-K on the 128 x 128 tap tile (small maps at batch 1-4: a handful of tiles, K in up to 32 splits): the splits meet INSIDE the
    // launch -- ticket per tile, the last arriver sums every split's slab in split order and runs the fused epilogue (conv_tap.hip) --
    // instead of in a splitk_reduce_kernel launch of its own (YOLOv3-608 at batch 1: 20 of 95 launches).
    static const bool no_inl = getenv("YOLO_NO_INLAUNCH_SPLITK") != nullptr;      // A/B switch (same results up to the fp32 summation order of the splits)
    // (up to eight splits: ONE workgroup reads them all -- 38 x 38 at batch 1, 2 splits: 23 -> 20 us; 19 x 19, 8 splits: 25.5 -> 24; beyond
    // that the reduce launch, which spreads the sum over the chip, wins: 13 x 13 float32 with 16 / 32 splits 38 -> 40.5 / 61 -> 67 us)
    if (ks > 1 && ks <= 8 && !pair && tile == 11 && !no_inl) {
        const long long mq = (long long)(p.M / p.HoWo) * (p.H + 1) * (p.W + 1);
        const long long nb11 = (mq + 127) / 128 * ((p.Cout + 127) / 128);
        if (nb11 * 128 <= (long long)kPairCounterBytes && (size_t)nb11 * (size_t)ks * 65536 <= slab_bytes) pair = 1;
    }
    return ConvPick{tile, ks, ku, pair};
}

// the shape fields pick_conv reads, for a batch, without device pointers (workspace sizing, kernel_info)
void conv_shape_params(const yolo_net *net, const Kernel &k, int batch, ConvParams &p) {
    memset(&p, 0, sizeof p);
    const yolo_layer_desc &d = net->layers[k.src_layer].d;
    p.H = k.in.H; p.W = k.in.W;
    p.Ho = net->layers[k.src_layer].H; p.Wo = net->layers[k.src_layer].W; p.HoWo = p.Ho * p.Wo;
    p.M = (int)((long long)batch * p.HoWo);
    p.Cout = k.cout;
    p.ksize = d.ksize; p.stride = d.stride; p.taps = d.ksize * d.ksize;
    p.ktiles = k.ktiles; p.cin_chunks = k.cpt;
    p.f32 = net->opt.dtype == YOLO_DTYPE_F32;
    p.out_f32 = k.out.f32 || p.f32;
    p.outmode = k.outmode; p.has_res = k.has_res;
    conv_vec_flags(net, k, p.out_f32 != 0, p.vec_out, p.vec_res);
    const long long ob = (long long)batch * k.out.img_stride * (p.out_f32 ? 4 : net->esize);
    p.out_bytes = ob > 0 && ob <= 0x7ffffff0LL ? (uint32_t)ob : 0u;
    const long long rb = k.has_res ? (long long)batch * k.in2.img_stride * net->esize : 0;
    p.res_bytes = rb > 0 && rb <= 0x7ffffff0LL ? (uint32_t)rb : 0u;
}

// float32 partial-sum slab one arena needs for ANY batch up to its share of max_batch (a net built for batch 32 also runs
// the short last batch of a TEST directory, where the small maps do split): 0 when no launch ever splits
size_t splitk_slab_bytes(const yolo_net *net) {
    const int per = net->arena_full ? net->opt.max_batch : (net->opt.max_batch + net->arenas - 1) / net->arenas;
    size_t need = 0;
    for (const Kernel &k : net->kernels) {
        if (k.kind != K_CONV || k.stem >= 2) continue;
        for (int b = 1; b <= per; ++b) {
            ConvParams p;
            conv_shape_params(net, k, b, p);
            const ConvPick pk = pick_conv(net, k, p, k.tile, kSplitkSlabMax);
            if (pk.pair) {
                const long long mq = (long long)(p.M / p.HoWo) * (p.H + 1) * (p.W + 1);
                const int nb = dma_cfg_nb(pk.tile);
                const long long ptiles = pk.tile == 22 ? (long long)(p.M / p.HoWo) : (mq + nb - 1) / nb;      // (22: a tile per image)
                const size_t bytes = (size_t)(ptiles * ((p.Cout + 127) / 128)) * (size_t)pk.ks * 128 * (size_t)nb * 4;
                if (bytes > need) need = bytes;
            } else if (pk.ks > 1) {
                const size_t bytes = (size_t)pk.ks * (size_t)p.M * (size_t)((p.Cout + 127) / 128 * 128) * 4;
                if (bytes > need) need = bytes;
            }
        }
    }
    // layout of an arena's slab: [ticket counters of the in-launch pair split, kPairCounterBytes | partial sums]: the counters must
    // never be written by anything but the pair kernels (they rely on finding them at zero)
    return need ? kPairCounterBytes + (need + 4095) / 4096 * 4096 : 0;
}

// Back-to-back 1x1: does the launch of conv `ki` at this batch also compute the 1x1 conv `ki + 1` (plan.cpp marked the pair)?
// Yes when the tile the batch picks holds all 128 couts of 256 positions per workgroup and has the fused instantiation: the 2-D
// 128 x 256 tap tile (12) or the 128 x 256 K32 LDS-DMA tile (6), whole K, lean epilogue.  On success `p` (the 3x3's launch
// parameters) carries the 1x1's weights, bias and output view.
bool conv_fuse2(const yolo_net *net, size_t ki, const Ptrs *P, int batch, ConvParams &p, size_t slab_bytes) {
    const Kernel &k = net->kernels[ki];
    if (!k.fuse2_next || ki + 1 >= net->kernels.size() || !net->kernels[ki + 1].fuse2_prev) return false;
    const ConvPick pk = pick_conv(net, k, p, k.tile, slab_bytes);
    if ((pk.tile != 12 && pk.tile != 6 && pk.tile != 23) || pk.ks > 1 || pk.pair || !conv_fast_epilogue_ok(p)) return false;
    if ((pk.tile == 12) != (p.has_res != 0)) return false;      // the instantiations built: 2-D tap tile + residual; LDS-DMA tile / stride-2 tap tile without
    const Kernel &b = net->kernels[ki + 1];
    const long long ob = (long long)batch * b.out.img_stride * net->esize;
    if (ob <= 0 || ob > 0x7ffffff0LL) return false;
    p.fuse2 = 1;
    p.w2 = weights_at(net, b.w_off);
    p.w2_bytes = (uint32_t)b.w_bytes;
    p.wrow2_bytes = (uint32_t)b.ktiles * 128;
    p.b2 = reinterpret_cast<const float *>(weights_at(net, b.b_off));
    p.out2 = P ? P->view_ptr(b.out) : nullptr;
    p.out2_bytes = (uint32_t)ob;
    p.out2_ld = b.out.ld;
    p.out2_img_stride = b.out.img_stride;
    p.leaky2 = b.leaky;
    return true;
}

hipError_t launch_conv_any(const yolo_net *net, const Kernel &k, const ConvParams &p0, int tile_req, hipStream_t s, int arena = 0) {
    const size_t slab = net->splitk_bytes / (size_t)net->arenas / 256 * 256;      // concurrent parts (streams) must not share a slab
    const size_t data_bytes = slab > kPairCounterBytes ? slab - kPairCounterBytes : 0;
    const ConvPick pk = pick_conv(net, k, p0, tile_req, data_bytes);
    const int tile = pk.tile, ks = pk.ks, ku = pk.ku;
    ConvParams p = p0;
    if (ks > 1) {
        p.ksplit = ks; p.kunits = ku;
        p.cout_pad = (p.Cout + 127) / 128 * 128;
        unsigned char *base = net->dev_ws + net->splitk_off + (size_t)arena * slab;
        p.part = reinterpret_cast<float *>(base + kPairCounterBytes);
        if (pk.pair) {      // counters (zeroed at bind, returned to zero by every launch) in front of the partial sums
            p.pair = 1;
            p.pair_cnt = reinterpret_cast<int *>(base);
            p.part_bytes = (uint32_t)(data_bytes < 0x7ffffff0u ? data_bytes : 0x7ffffff0u);
        }
    }
    if (tile <= 0) p.f32_emu = conv_f32_emu_rule(net->opt.f32_products, net->opt.dtype, p, k.cfg, k.perchunk != 0, ks) ? 1 : 0;
    hipError_t e = tile > 0 ? launch_conv_dma(p, tile, s) : launch_conv(p, net->opt.dtype, k.cfg, k.perchunk != 0, s);
    if (e != hipSuccess || ks <= 1 || pk.pair) return e;
    ReduceParams r;
    memset(&r, 0, sizeof r);
    r.part = p.part; r.bias = p.bias; r.res = p.has_res ? p.res : nullptr; r.out = p.out;
    r.obj_out = p.obj_out; r.obj_width = p.obj_width; r.obj_rows = p.obj_rows; r.obj_row0 = p.obj_row0; r.obj_na = p.obj_na;
    r.ksplit = ks; r.M = p.M; r.Cout = p.Cout; r.cout_pad = p.cout_pad; r.HoWo = p.HoWo; r.Wo = p.Wo;
    r.out_ld = p.out_ld; r.res_ld = p.res_ld; r.leaky = p.leaky; r.outmode = p.outmode; r.out_f32 = p.out_f32; r.f32 = p.f32;
    r.out_img_stride = p.out_img_stride; r.res_img_stride = p.res_img_stride;
    return launch_splitk_reduce(r, s);
}

// May the branch tails of this net run beside its main chain at this batch?  Yes when the plan has any and no launch of the pass splits K.
bool branch_tails_ok(yolo_net *net, int batch) {
    static const bool off = getenv("YOLO_NO_BRANCH_STREAM") != nullptr;       // A/B switch (same results either way)
    if (off || net->side_chains <= 0 || net->opt.keep_all || batch <= 0 || batch > net->opt.max_batch) return false;
    if (net->side_ok.size() != (size_t)net->opt.max_batch + 1) net->side_ok.assign((size_t)net->opt.max_batch + 1, -1);
    signed char &memo = net->side_ok[(size_t)batch];
    if (memo < 0) {
        const size_t slab = net->splitk_bytes / (size_t)net->arenas / 256 * 256;
        const size_t data_bytes = slab > kPairCounterBytes ? slab - kPairCounterBytes : 0;
        memo = 1;
        for (const Kernel &k : net->kernels) {
            if (k.kind != K_CONV || k.stem >= 2) continue;
            ConvParams p;
            conv_shape_params(net, k, batch, p);
            if (pick_conv(net, k, p, k.tile, data_bytes).ks > 1) { memo = 0; break; }
        }
    }
    return memo == 1;
}
int branch_streams(yolo_net *net, int arena) {
    if (net->branch.empty()) {
        const size_t n = (size_t)(net->arenas > 0 ? net->arenas : 1);
        net->branch.assign(n, nullptr);
        net->e_bjoin.assign(n, nullptr);
        net->e_bfork.assign(n * 4, nullptr);
        for (size_t i = 0; i < n; ++i)
            if (hipStreamCreateWithFlags(&net->branch[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&net->e_bjoin[i], hipEventDisableTiming) != hipSuccess)
                return fail(YOLO_ERR_HIP, "branch tail: stream / event creation failed");
        for (size_t i = 0; i < n * 4; ++i)
            if (hipEventCreateWithFlags(&net->e_bfork[i], hipEventDisableTiming) != hipSuccess)
                return fail(YOLO_ERR_HIP, "branch tail: event creation failed");
    }
    return (size_t)arena < net->branch.size() ? YOLO_OK : fail(YOLO_ERR_STATE, "branch tail: arena out of range");
}

int run_forward_pass(yolo_net *net, const float *in_dev, int batch, float *out_dev, hipStream_t s, hipEvent_t *ev, int img0, int arena,
                     long long *obj_rows_out) {
    Ptrs P{net, in_dev, out_dev, img0, arena};
    const int dtype = net->opt.dtype;
    const int epc = net->epc;
    long long obj_rows_written = 0;          // rows of the compact objectness array the head convs of this pass fill
    bool fused2_done = false;                // the previous conv launch has computed this 1x1 conv too (back-to-back fusion)
    // branch tails (plan.cpp: side_chains) on a second stream of this part: fork by an event behind the kernel in front of the tail,
    // one join in front of whatever follows the pass (the decode, the caller).  Not under per-kernel events, and not at a batch where
    // any launch of the pass splits K (the split-K slab and its ticket counters are one per arena)
    const bool use_branch = !ev && branch_tails_ok(net, batch);
    hipStream_t const s_main = s;
    hipStream_t s_branch = nullptr;
    int cur_tail = 0;
    if (use_branch) {
        const int rc = branch_streams(net, arena);
        if (rc) return rc;
        s_branch = net->branch[arena];
    }
    for (size_t ki = 0; ki < net->kernels.size(); ++ki) {
        const Kernel &k = net->kernels[ki];
        hipError_t e = hipSuccess;
        s = s_main;
        if (use_branch && k.side) {
            if (k.side != cur_tail) {
                hipEvent_t ef = net->e_bfork[(size_t)arena * 4 + (size_t)((k.side - 1) & 3)];
                if (hipEventRecord(ef, s_main) != hipSuccess || hipStreamWaitEvent(s_branch, ef, 0) != hipSuccess)
                    return fail(YOLO_ERR_HIP, "branch tail: fork failed");
                cur_tail = k.side;
            }
            s = s_branch;
        }
        if (ev && hipEventRecord(ev[2 * ki], s) != hipSuccess) return fail(YOLO_ERR_HIP, "hipEventRecord failed");
        switch (k.kind) {
        case K_PREP: {
            PrepParams p;
            p.in = in_dev;
            p.out = P.view_ptr(k.out);
            p.pixels = (long long)batch * k.in.H * k.in.W;
            p.C = k.in.C;
            p.Cpad = k.out.ld;
            e = launch_prep(p, dtype, s);
            break;
        }
        case K_CONV: {
            if (k.stem == 2) {      // the previous kernel (first layer) is folded into this launch
                const Kernel &f = net->kernels[ki - 1];
                StemParams p;
                memset(&p, 0, sizeof p);
                p.in = in_dev;
                p.w1 = reinterpret_cast<const float *>(net->dev_weights + f.w_off);
                p.b1 = reinterpret_cast<const float *>(net->dev_weights + f.b_off);
                p.w2 = net->dev_weights + k.w_off;
                p.b2 = reinterpret_cast<const float *>(net->dev_weights + k.b_off);
                p.w2_bytes = (uint32_t)k.w_bytes;
                p.wrow2 = (uint32_t)k.ktiles * 128;
                p.out = P.view_ptr(k.out);
                p.H = f.in.H; p.W = f.in.W; p.Ho = k.out.H; p.Wo = k.out.W;
                p.out_ld = k.out.ld; p.out_img_stride = k.out.img_stride;
                p.in_img_stride = (long long)f.in.H * f.in.W * 3;
                if (ki + 1 < net->kernels.size() && net->kernels[ki + 1].stem == 3) {      // 1x1 64->32 on the same pixels
                    const Kernel &t = net->kernels[ki + 1];
                    p.w3 = net->dev_weights + t.w_off;
                    p.b3 = reinterpret_cast<const float *>(net->dev_weights + t.b_off);
                    p.out3 = P.view_ptr(t.out);
                    p.out3_ld = t.out.ld; p.out3_img_stride = t.out.img_stride;
                }
                e = launch_stem(p, batch, s, net->halves ? 512 / net->parts : 512);
                break;
            }
            if (k.stem == 3) break;     // computed by the stem kernel
            if (k.fuse2_prev && fused2_done) { fused2_done = false; break; }       // computed by the conv in front of it
            ConvParams p;
            int rc = make_conv_params(net, k, P, batch, p);
            if (rc) return rc;
            if (k.head && p.obj_out) obj_rows_written += (long long)p.Ho * p.Wo * p.obj_na;
            {
                const size_t slab = net->splitk_bytes / (size_t)net->arenas / 256 * 256;
                fused2_done = conv_fuse2(net, ki, &P, batch, p, slab > kPairCounterBytes ? slab - kPairCounterBytes : 0);
            }
            e = launch_conv_any(net, k, p, k.tile, s, P.arena);
            break;
        }
        case K_FIRST: {
            if (k.stem == 1) break;     // runs inside the next kernel (stem.hip)
            FirstParams p;
            p.in = in_dev;
            p.wgt = reinterpret_cast<const float *>(net->dev_weights + k.w_off);
            p.bias = reinterpret_cast<const float *>(net->dev_weights + k.b_off);
            p.out = P.view_ptr(k.out);
            p.H = k.in.H; p.W = k.in.W; p.Cout = k.cout; p.out_ld = k.out.ld; p.leaky = k.leaky;
            p.pool = k.pool_fused;
            p.round_half = dtype == YOLO_DTYPE_F16;
            p.out_img_stride = k.out.img_stride;
            p.total = (long long)batch * k.in.H * k.in.W;
            if (k.out.ld % epc || (k.out.base + k.out.coff) % epc || k.out.img_stride % epc)
                return fail(YOLO_ERR_PLAN, "first-layer kernel needs a 16-byte aligned output view");
            e = launch_first(p, dtype, s);
            break;
        }
        case K_POOL: {
            PoolParams p;
            p.in = P.view_ptr(k.in);
            p.out = P.view_ptr(k.out);
            p.H = k.in.H; p.W = k.in.W; p.C = k.in.C; p.in_ld = k.in.ld;
            p.Ho = k.out.H; p.Wo = k.out.W; p.out_ld = k.out.ld; p.stride = k.pool_stride;
            p.in_img_stride = k.in.img_stride; p.out_img_stride = k.out.img_stride;
            p.total = (long long)batch * k.out.H * k.out.W;
            e = launch_pool(p, dtype, s);
            break;
        }
        case K_ELTWISE: {
            EltParams p;
            memset(&p, 0, sizeof p);
            p.a = P.view_ptr(k.in);
            p.a_f32 = k.in.f32;
            p.b = k.has_res ? P.view_ptr(k.in2) : nullptr;
            p.out = P.view_ptr(k.out);
            p.H = k.in.H; p.W = k.in.W; p.C = k.in.C;
            p.a_ld = k.in.ld; p.b_ld = k.in2.ld; p.out_ld = k.out.ld;
            p.outmode = k.outmode;
            p.out_f32 = k.out.f32 || dtype == YOLO_DTYPE_F32;
            p.a_img_stride = k.in.img_stride; p.b_img_stride = k.in2.img_stride; p.out_img_stride = k.out.img_stride;
            p.total = (long long)batch * k.in.H * k.in.W * k.in.C;
            e = launch_eltwise(p, dtype, s);
            break;
        }
        }
        if (e != hipSuccess) {
            char msg[160];
            snprintf(msg, sizeof msg, "kernel %zu (layer %d) launch failed: %s", ki, k.layer, hipGetErrorString(e));
            return fail(YOLO_ERR_HIP, msg);
        }
        if (ev && hipEventRecord(ev[2 * ki + 1], s) != hipSuccess) return fail(YOLO_ERR_HIP, "hipEventRecord failed");
    }
    if (cur_tail) {
        if (hipEventRecord(net->e_bjoin[arena], s_branch) != hipSuccess || hipStreamWaitEvent(s_main, net->e_bjoin[arena], 0) != hipSuccess)
            return fail(YOLO_ERR_HIP, "branch tail: join failed");
    }
    *obj_rows_out = obj_rows_written;
    return YOLO_OK;
}

// One forward pass; with YOLO_STREAMS=N (N = 2..4) the batch goes out as N parts on N streams (the caller's and internal
// ones, fork/join by events): images are independent, so the ragged tail + cold start of every kernel of one part overlaps
// the bulk of the other parts' kernels instead of leaving CUs idle at each of the ~73 kernel boundaries.  Every part has
// its own activation arena (plan.cpp: allocate).
// every head conv of the plan fills the compact objectness array at this batch (what run_forward_impl finds out afterwards as obj_valid)
bool all_heads_write_objectness(yolo_net *net, const float *in_dev, float *out_dev, int batch) {
    if (!net->obj_bytes || net->head.n_classes <= 0) return false;
    const int per = net->parts >= 2 && batch > part_batch(net) ? part_batch(net) : batch;
    Ptrs P{net, in_dev, out_dev, 0, 0};
    bool any = false;
    for (const Kernel &k : net->kernels) {
        if (k.kind != K_CONV || !k.head) continue;
        ConvParams p;
        if (make_conv_params(net, k, P, per, p) != 0 || !p.obj_out) return false;
        // (the rows reach the compact array through the staged float32 epilogue of the LDS-DMA tiles or the 4-wave kernel, or through
        // the split-K reduce kernel: all of them honour obj_out)
        any = true;
    }
    return any;
}

int run_forward_impl(yolo_net *net, const float *in_dev, int batch, float *out_dev, hipStream_t s, hipEvent_t *ev);
int run_forward(yolo_net *net, const float *in_dev, int batch, float *out_dev, hipStream_t s, hipEvent_t *ev = nullptr) {
    const int rc = run_forward_impl(net, in_dev, batch, out_dev, s, ev);
    if (rc != YOLO_OK) {        // leave the pair-split counters as every later launch expects them; keep the first error's message
        const std::string msg = get_error();
        (void)hipGetLastError();
        (void)zero_pair_counters(net);
        set_error(msg);
    }
    return rc;
}
int run_forward_impl(yolo_net *net, const float *in_dev, int batch, float *out_dev, hipStream_t s, hipEvent_t *ev) {
    const long long rows = net->head.n_classes > 0 ? (long long)(net->out_count / (size_t)(5 + net->head.n_classes)) : -1;
    const int parts = net->parts;
    const int per = (net->opt.max_batch + parts - 1) / parts;       // images a part holds
    if (parts >= 2 && batch > per) {
        if (ev) return fail(YOLO_ERR_STATE, "per-kernel events need the parts timed one by one (yolo_net_forward_timed)");
        if (net->side.empty()) {
            net->side.resize(parts - 1);
            net->e_join.resize(parts - 1);
            for (int i = 0; i < parts - 1; ++i)
                if (hipStreamCreateWithFlags(&net->side[i], hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&net->e_join[i], hipEventDisableTiming) != hipSuccess)
                    return fail(YOLO_ERR_HIP, "multi-stream forward: stream/event creation failed");
            if (hipEventCreateWithFlags(&net->e_fork, hipEventDisableTiming) != hipSuccess)
                return fail(YOLO_ERR_HIP, "multi-stream forward: event creation failed");
        }
        const yolo_layer_desc &d0 = net->layers[0].d;
        const size_t in_img = (size_t)d0.h * d0.w * d0.c;
        net->halves = true;         // persistent kernels size their grids for a share of the chip
        HIP_TRY(hipEventRecord(net->e_fork, s));
        bool all = true;
        int used = 0;
        for (int part = 0, img0 = 0; img0 < batch; ++part, img0 += per) {
            const int nb = batch - img0 < per ? batch - img0 : per;
            hipStream_t st = part == 0 ? s : net->side[part - 1];
            if (part > 0) HIP_TRY(hipStreamWaitEvent(st, net->e_fork, 0));
            long long w = 0;
            const int rc = run_forward_pass(net, in_dev + (size_t)img0 * in_img, nb, out_dev + (size_t)img0 * net->out_count, st, nullptr,
                                            img0, part, &w);
            if (rc) return rc;
            all = all && w == rows;
            used = part + 1;
        }
        for (int part = 1; part < used; ++part) {
            HIP_TRY(hipEventRecord(net->e_join[part - 1], net->side[part - 1]));
            HIP_TRY(hipStreamWaitEvent(s, net->e_join[part - 1], 0));
        }
        net->obj_valid = net->obj_bytes > 0 && rows > 0 && all;
        return YOLO_OK;
    }
    net->halves = false;
    long long w0 = 0;
    int rc = run_forward_pass(net, in_dev, batch, out_dev, s, ev, 0, 0, &w0);
    if (rc) return rc;
    net->obj_valid = net->obj_bytes > 0 && rows > 0 && w0 == rows;
    return YOLO_OK;
}

int check_ready(yolo_net *net, const void *in, int batch, const char *who) {
    if (!net || !in) return fail(YOLO_ERR_ARG, std::string(who) + ": null argument");
    if (batch <= 0 || batch > net->opt.max_batch) return fail(YOLO_ERR_ARG, std::string(who) + ": batch outside 1..max_batch");
    if (!net->weights_loaded && net->weight_count) return fail(YOLO_ERR_STATE, std::string(who) + ": weights not loaded");
    if (!net->dev_ws) return fail(YOLO_ERR_STATE, std::string(who) + ": workspace not bound");
    return YOLO_OK;
}

void fill_decode(const yolo_head_desc &h, DecodeParams &dp) {
    dp.version = h.version;
    dp.n_classes = h.n_classes;
    dp.n_scales = h.n_scales;
    int row0 = 0;
    for (int s = 0; s < h.n_scales; ++s) {
        dp.sc[s].row0 = row0; dp.sc[s].h = h.h[s]; dp.sc[s].w = h.w[s]; dp.sc[s].na = h.n_anchors[s];
        for (int a = 0; a < h.n_anchors[s]; ++a) { dp.sc[s].aw[a] = h.anchors[s][2 * a]; dp.sc[s].ah[a] = h.anchors[s][2 * a + 1]; }
        row0 += h.h[s] * h.w[s] * h.n_anchors[s];
    }
    dp.rows = row0;
}

int run_decode_nms(const yolo_head_desc &h, const float *logits, int batch, double thr, double iou, int mode, int cap,
                   int max_boxes, unsigned char *cand, int *cand_count, yolo_box *boxes, int32_t *counts, int32_t *status,
                   int32_t *keep_idx, hipStream_t s, unsigned char *nms_scratch = nullptr, const float *obj = nullptr, int *counters_clean = nullptr) {
    DecodeParams dp;
    memset(&dp, 0, sizeof dp);
    fill_decode(h, dp);
    dp.logits = logits;
    dp.obj = obj;
    dp.threshold = (float)thr;      // `p < threshold` compares in float32 under NumPy 2 (weak Python float)
    dp.cap = cap;
    dp.cand = cand;
    dp.cand_count = cand_count;
    dp.total_rows = (long long)batch * dp.rows;
    // counters_clean (yolo_net_detect): the counters are the net's own; the NMS kernel returns each counter it read to zero, so only the
    // first detect, one behind a failed one, and one at a LARGER batch than any before it need the memset launch (4.5 us: half a
    // percent of a batch-1 step).  *counters_clean = how many counters, from the first, are known to be zero: the memset and the NMS
    // reset cover the current call's `batch` counters only (a detect of 1 image followed by one of 32 found counters 1..31 as the
    // caller's workspace held them -- uninitialised memory for a C-ABI caller: ADVICE r4).
    const int clean_n = counters_clean ? *counters_clean : 0;
    const bool need_clear = batch > clean_n;
    if (counters_clean) *counters_clean = 0;
    HIP_TRY(launch_decode(dp, batch, s, need_clear));
    NmsParams np;
    np.cand = reinterpret_cast<const Candidate *>(cand);
    np.cand_count = cand_count;
    np.cap = cap; np.max_boxes = max_boxes; np.mode = mode;
    np.iou_threshold = iou;
    np.boxes = boxes; np.counts = counts; np.status = status; np.keep_idx = keep_idx;
    np.scratch = nms_scratch; np.scratch_stride = nms_scratch_bytes(cap);
    np.reset_count = counters_clean ? cand_count : nullptr;
    HIP_TRY(launch_nms(np, batch, s));
    if (counters_clean) *counters_clean = need_clear ? batch : clean_n;
    return YOLO_OK;
}

}  // namespace

extern "C" {

int yolo_net_forward(yolo_net *net, const float *in_dev, int batch, float *out_dev, void *stream) {
    int rc = check_ready(net, in_dev, batch, "yolo_net_forward");
    if (rc) return rc;
    if (!out_dev) return fail(YOLO_ERR_ARG, "yolo_net_forward: null output");
    return run_forward(net, in_dev, batch, out_dev, static_cast<hipStream_t>(stream));
}

int yolo_net_forward_timed(yolo_net *net, const float *in_dev, int batch, float *out_dev, void *stream, float *ms_host) {
    int rc = check_ready(net, in_dev, batch, "yolo_net_forward_timed");
    if (rc) return rc;
    if (!out_dev || !ms_host) return fail(YOLO_ERR_ARG, "yolo_net_forward_timed: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nk = net->kernels.size();
    std::vector<hipEvent_t> ev(2 * nk, nullptr);
    for (auto &e : ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(YOLO_ERR_HIP, "hipEventCreate failed");
    // with several arenas the parts are timed one after the other on the caller's stream (each kernel alone on the chip)
    // and their times added per kernel
    const int parts = net->parts;
    const int per = (net->opt.max_batch + parts - 1) / parts;
    const yolo_layer_desc &d0 = net->layers[0].d;
    const size_t in_img = (size_t)d0.h * d0.w * d0.c;
    for (size_t k = 0; k < nk; ++k) ms_host[k] = 0.f;
    net->halves = false;
    long long rows_written = 0;
    for (int part = 0, img0 = 0; rc == YOLO_OK && img0 < batch; ++part, img0 += per) {
        const int nb = batch - img0 < per ? batch - img0 : per;
        long long w = 0;
        rc = run_forward_pass(net, in_dev + (size_t)img0 * in_img, nb, out_dev + (size_t)img0 * net->out_count, s, ev.data(), img0, part, &w);
        rows_written += w;
        if (rc == YOLO_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(YOLO_ERR_HIP, "hipStreamSynchronize failed");
        for (size_t k = 0; rc == YOLO_OK && k < nk; ++k) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]) != hipSuccess) rc = fail(YOLO_ERR_HIP, "hipEventElapsedTime failed");
            ms_host[k] += t;
        }
    }
    net->obj_valid = false;     // (the timed pass is not followed by a decode)
    (void)rows_written;
    if (rc != YOLO_OK) { const std::string msg = get_error(); (void)hipGetLastError(); (void)zero_pair_counters(net); set_error(msg); }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
}

int yolo_net_kernel_info(const yolo_net *net, int kernel, yolo_kernel_info *out) {
    if (!net || !out || kernel < 0 || kernel >= (int)net->kernels.size()) return fail(YOLO_ERR_ARG, "yolo_net_kernel_info: bad argument");
    const Kernel &k = net->kernels[kernel];
    memset(out, 0, sizeof *out);
    out->kind = k.kind; out->layer = k.layer;
    auto set_symbol = [&](const std::string &sym) { snprintf(out->symbol, sizeof out->symbol, "%s", sym.c_str()); };
    const char *t = net->opt.dtype == YOLO_DTYPE_F16 ? "f16" : "f32";
    auto elems = [](const View &v) { return (double)v.H * v.W * v.C; };
    auto esz = [&](const View &v) { return v.f32 ? 4.0 : (double)net->esize; };
    if (k.kind == K_CONV) {
        const LayerInfo &li = net->layers[k.src_layer];
        out->variant = k.cfg + 4 * k.perchunk;
        out->ksize = k.ksize; out->stride = k.stride; out->cin = k.cin; out->cout = k.cout; out->out_h = li.H; out->out_w = li.W;
        out->flops = 2.0 * li.H * li.W * k.cout * k.ksize * k.ksize * k.cin;
        out->bytes = (double)k.in.H * k.in.W * k.cin * net->esize + elems(k.out) * esz(k.out) + (k.has_res ? elems(k.in2) * net->esize : 0.0);
        out->weight_bytes = (double)k.cout * k.ksize * k.ksize * k.cin * net->esize + 4.0 * k.cout;
        if (k.stem == 3) {          // no launch of its own
            out->flops = 0; out->bytes = 0; out->weight_bytes = 0;
            snprintf(out->name, sizeof out->name, "conv_igemm<fused into conv_stem>");
            return YOLO_OK;
        }
        if (k.stem == 2) {
            const Kernel &f = net->kernels[kernel - 1];
            out->flops += 2.0 * f.out.H * f.out.W * f.cout * 27;
            out->bytes = (double)f.in.H * f.in.W * 3 * 4 + elems(k.out) * esz(k.out);
            out->weight_bytes += 28.0 * f.cout * 4;
            snprintf(out->name, sizeof out->name, "conv_stem<f16,3-32-64>");
            set_symbol("yolo::stem_v3_kernel(yolo::StemParams)");
            if (kernel + 1 < (int)net->kernels.size() && net->kernels[kernel + 1].stem == 3) {
                const Kernel &t3 = net->kernels[kernel + 1];
                out->flops += 2.0 * li.H * li.W * t3.cout * t3.cin;
                out->bytes += elems(t3.out) * esz(t3.out);
                out->weight_bytes += (double)t3.cout * t3.cin * net->esize + 4.0 * t3.cout;
                snprintf(out->name, sizeof out->name, "conv_stem<f16,3-32-64-32>");
            }
            return YOLO_OK;
        }
        // which kernel runs at max_batch (bench.py runs at max_batch): the same decision the launch path takes
        ConvParams sp;
        const int per_arena = part_batch(net);
        conv_shape_params(net, k, per_arena, sp);
        const size_t slab_i = net->splitk_bytes / (size_t)net->arenas / 256 * 256;
        const size_t slab_d = slab_i > kPairCounterBytes ? slab_i - kPairCounterBytes : 0;
        if (k.fuse2_prev && kernel > 0) {       // computed by the conv in front of it at this batch?
            ConvParams pp;
            conv_shape_params(net, net->kernels[kernel - 1], per_arena, pp);
            if (conv_fuse2(net, (size_t)kernel - 1, nullptr, per_arena, pp, slab_d)) {
                out->flops = 0; out->bytes = 0; out->weight_bytes = 0;
                snprintf(out->name, sizeof out->name, "conv_igemm<fused into the conv in front>");
                return YOLO_OK;
            }
        }
        const bool fused2 = conv_fuse2(net, (size_t)kernel, nullptr, per_arena, sp, slab_d);
        if (fused2) {       // this launch also computes the 1x1 behind it: its work and its output belong here
            const Kernel &b2 = net->kernels[kernel + 1];
            out->flops += 2.0 * li.H * li.W * b2.cout * b2.cin;
            out->bytes += elems(b2.out) * esz(b2.out);
            out->weight_bytes += (double)b2.cout * b2.cin * net->esize + 4.0 * b2.cout;
        }
        const ConvPick pk = pick_conv(net, k, sp, k.tile, slab_d);
        const int tile = pk.tile;
        const bool f32net = net->opt.dtype == YOLO_DTYPE_F32;
        if (tile > 0) {
            out->variant = 8 + tile;
            snprintf(out->name, sizeof out->name, "conv_igemm_dma<%s,%s>", t, dma_cfg_name(tile));
            sp.ksplit = pk.ks;
            std::string sym = dma_cfg_symbol_for(tile, f32net, sp);      // (the persistent form of the tap kernel where it takes the launch)
            if (pk.ks > 1) {        // the split-K instantiation of the tap kernel (its last template argument)
                const size_t at = sym.rfind(", false, false, false>(");
                if (at != std::string::npos) sym.replace(at, 23, ", true, false, false>(");
                const size_t occ = sym.find("26, 4, 1, true, false, false>(");       // the in-launch pair on the 128 x 256 tile is built for one workgroup per CU
                if (pk.pair && occ != std::string::npos) sym.replace(occ, 30, "26, 2, 1, true, false, false>(");
                const size_t occ22 = sym.find("14, 4, 1, true, false, false>(");     // ... and on the image-aligned 128 x 192 tile
                if (pk.pair && occ22 != std::string::npos) sym.replace(occ22, 30, "14, 2, 1, true, false, false>(");
            }
            set_symbol(sym);
        } else {
            const bool emu = conv_f32_emu_rule(net->opt.f32_products, net->opt.dtype, sp, k.cfg, k.perchunk != 0, pk.ks);
            set_symbol(conv_symbol(net->opt.dtype, k.cfg, k.perchunk != 0, emu));
            if (emu) snprintf(out->name, sizeof out->name, "conv_igemm_emu<f32 as 9 x bf16,N128>");
            else snprintf(out->name, sizeof out->name, "conv_igemm<%s,N%d,%s>", t, k.cfg == CFG_N128 ? 128 : k.cfg == CFG_N64 ? 64 : 32,
                          k.perchunk ? "perchunk" : "uniform");
        }
        if (k.outmode == OUT_POOL2) {
            const size_t n = strlen(out->name);
            snprintf(out->name + n, sizeof out->name - n, "+pool");
        }
        if (fused2) {
            const size_t n = strlen(out->name);
            snprintf(out->name + n, sizeof out->name - n, "+1x1");
        }
        if (pk.pair) {              // K in two halves (or pk.ks splits) inside the launch
            const size_t n = strlen(out->name);
            if (pk.ks == 2) snprintf(out->name + n, sizeof out->name - n, "+pairK");
            else snprintf(out->name + n, sizeof out->name - n, "+splitK%d,1launch", pk.ks);
            out->bytes += (double)pk.ks * (double)li.H * li.W * ((k.cout + 127) / 128 * 128) * 4.0;
        } else if (pk.ks > 1) {     // two launches: K splits into the float32 slab, then splitk_reduce_kernel (sum + fused epilogue)
            const size_t n = strlen(out->name);
            snprintf(out->name + n, sizeof out->name - n, "+splitK%d", pk.ks);
            out->bytes += 2.0 * pk.ks * (double)li.H * li.W * ((k.cout + 127) / 128 * 128) * 4.0;      // partial sums written + read once
        }
    } else if (k.kind == K_FIRST) {
        out->ksize = 3; out->stride = 1; out->cin = 3; out->cout = k.cout; out->out_h = k.out.H; out->out_w = k.out.W;
        out->flops = 2.0 * k.out.H * k.out.W * k.cout * 27;
        out->bytes = (double)k.in.H * k.in.W * 3 * 4 + elems(k.out) * esz(k.out);
        out->weight_bytes = 28.0 * k.cout * 4;
        snprintf(out->name, sizeof out->name, k.pool_fused ? "conv_first_pool<%s,%d>" : "conv_first<%s,%d>", t, k.cout);
        if (k.pool_fused) { out->out_h = k.in.H; out->out_w = k.in.W; out->flops = 2.0 * k.in.H * k.in.W * k.cout * 27; }
        set_symbol(first_symbol(net->opt.dtype, k.cout, k.pool_fused != 0));
        if (k.stem == 1) {
            out->symbol[0] = 0;          // no launch of its own: accounted for in the conv_stem kernel that follows
            out->flops = 0; out->bytes = 0; out->weight_bytes = 0;
            snprintf(out->name, sizeof out->name, "conv_first<fused into conv_stem>");
        }
    } else {
        out->out_h = k.out.H; out->out_w = k.out.W; out->cout = k.out.C; out->cin = k.in.C;
        out->bytes = elems(k.in) * (k.in.f32 ? 4.0 : net->esize) + elems(k.out) * esz(k.out) + (k.has_res ? elems(k.in2) * net->esize : 0.0);
        snprintf(out->name, sizeof out->name, "%s<%s>", k.kind == K_PREP ? "prep" : k.kind == K_POOL ? "pool" : "eltwise", t);
        const int epc = net->epc;       // pool: the 16-byte-vector instantiation runs when every stride is chunk-aligned (aux.hip)
        const bool vec = k.kind == K_POOL && k.in.C % epc == 0 && k.in.ld % epc == 0 && k.out.ld % epc == 0 && (k.in.base + k.in.coff) % epc == 0 &&
                         (k.out.base + k.out.coff) % epc == 0 && k.in.img_stride % epc == 0 && k.out.img_stride % epc == 0;
        set_symbol(aux_symbol(k.kind, net->opt.dtype, vec));
    }
    return YOLO_OK;
}

int yolo_net_tune_streams(yolo_net *net, const float *in_dev, int batch, void *stream) {
    int rc = check_ready(net, in_dev, batch, "yolo_net_tune_streams");
    if (rc) return rc;
    if (!net->arena_full || batch <= (net->opt.max_batch + 1) / 2) return YOLO_OK;       // nothing to choose (or not with this batch)
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *logits = reinterpret_cast<float *>(net->dev_ws + net->logits_off);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(YOLO_ERR_HIP, "yolo_net_tune_streams: hipEventCreate failed");
    }
    // interleaved: one pass, two halves, one pass, ... -- three forward passes per sample, the first round of each only warms up,
    // the best of the other four counts (boxes differ: the same build gains 3-4 % from two halves on one MI355X and loses 1-2 % on
    // another, so the rule's answer is re-measured where the net runs)
    float best[3] = {0.f, 1e30f, 1e30f};
    for (int rep = 0; rep < 5 && rc == YOLO_OK; ++rep)
        for (int parts = 1; parts <= 2 && rc == YOLO_OK; ++parts) {
            net->parts = parts;
            if (hipEventRecord(e0, s) != hipSuccess) rc = fail(YOLO_ERR_HIP, "yolo_net_tune_streams: hipEventRecord failed");
            for (int k = 0; k < 3 && rc == YOLO_OK; ++k) rc = run_forward(net, in_dev, batch, logits, s);
            if (rc == YOLO_OK && (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess))
                rc = fail(YOLO_ERR_HIP, "yolo_net_tune_streams: event failed");
            float ms = 0.f;
            if (rc == YOLO_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && rep > 0 && ms < best[parts]) best[parts] = ms;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    // two halves must win by 1.5 %: where they do not gain 3-4 % they are within +-1 % of one pass, which the later steady state
    // (warmer chip, decode + NMS behind the join) has been seen to turn into a loss
    net->parts = (rc == YOLO_OK && best[2] < 0.985f * best[1]) ? 2 : (rc == YOLO_OK ? 1 : net->arenas);
    net->parts_tuned = rc == YOLO_OK;
    net->obj_valid = false;
    return rc;
}

int yolo_net_set_streams(yolo_net *net, int parts) {
    if (!net) return fail(YOLO_ERR_ARG, "yolo_net_set_streams: null net");
    if (parts == net->parts) return YOLO_OK;
    if (!net->arena_full || parts < 1 || parts > net->arenas)
        return fail(YOLO_ERR_STATE, "yolo_net_set_streams: this net was planned for " + std::to_string(net->parts) + " part(s) only (streams given explicitly, or the rule says one)");
    net->parts = parts;
    net->parts_tuned = true;
    net->obj_valid = false;
    return YOLO_OK;
}

int yolo_net_autotune(yolo_net *net, const float *in_dev, int batch, void *stream) {
    int rc = check_ready(net, in_dev, batch, "yolo_net_autotune");
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *logits = reinterpret_cast<float *>(net->dev_ws + net->logits_off);
    {   // every launch of a multi-stream net sees one part of the batch: tune for that size (arena 0)
        const int per = part_batch(net);
        if (batch > per) batch = per;
    }
    rc = run_forward(net, in_dev, batch, logits, s);      // real activations in every buffer
    if (rc) return rc;
    Ptrs P{net, in_dev, logits};
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    for (Kernel &k : net->kernels) {
        if (k.kind != K_CONV || k.stem >= 2 || !dma_eligible(net, k)) continue;
        ConvParams p;
        rc = make_conv_params(net, k, P, batch, p);
        if (rc) break;
        float best = 1e30f;
        int best_tile = -1;
        for (int tile = 0; tile < dma_num_cfgs(); ++tile) {
            if (!conv_tile_valid(net, k, tile)) continue;
            float ms = 1e30f;
            bool ok = true;
            for (int rep = 0; rep < 4 && ok; ++rep) {       // first launch warms caches; keep the best of the rest
                ok = hipEventRecord(e0, s) == hipSuccess && launch_conv_any(net, k, p, tile, s) == hipSuccess &&
                     hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
                float t = 0.f;
                if (ok && rep > 0 && hipEventElapsedTime(&t, e0, e1) == hipSuccess && t < ms) ms = t;
            }
            if (!ok) { rc = fail(YOLO_ERR_HIP, "yolo_net_autotune: launch failed"); break; }
            if (ms < best) { best = ms; best_tile = tile; }
        }
        if (rc) break;
        k.tile = best_tile;
        net->side_ok.clear();       // (whether a pass splits K -- branch_tails_ok -- depends on the tiles)
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int yolo_net_detect(yolo_net *net, const float *in_dev, int batch, double threshold, double iou_threshold, int nms_mode,
                    yolo_box *boxes_dev, int32_t *counts_dev, int32_t *status_dev, void *stream) {
    int rc = check_ready(net, in_dev, batch, "yolo_net_detect");
    if (rc) return rc;
    if (!boxes_dev || !counts_dev || !status_dev) return fail(YOLO_ERR_ARG, "yolo_net_detect: null output");
    std::string err;
    rc = check_head(&net->head, net->out_count, err);
    if (rc) return fail(YOLO_ERR_STATE, "yolo_net_detect: head geometry not set (" + err + "); call yolo_net_set_head");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *logits = reinterpret_cast<float *>(net->dev_ws + net->logits_off);
    // The logits of a detect call never leave the workspace, and the decode kernel reads the objectness of every row from the compact
    // array the head convs fill and the rest of a row only where sigmoid(objectness) reaches the threshold (p = sigmoid(obj) in v3,
    // <= sigmoid(obj) in v2): rows 0.01 below the threshold's logit are not written (conv_common.h: conv_epilogue_f32_staged).  Only
    // when EVERY head conv of this plan writes the compact array -- else the decode would look for objectness in rows never written.
    static const bool dense = getenv("YOLO_DENSE_LOGITS") != nullptr;       // A/B switch (same boxes either way)
    net->obj_min_logit = -INFINITY;
    // (thresholds within 1e-4 of 1 run dense: there the decode's float32 sigmoid saturates -- sigmoid_f32(x) rounds to 1 - 2^-23 over a range
    // of x wider than the 0.01 margin, so rows the decode accepts would lie below the cut; at 1 - 1e-4 the margin is still ten float32
    // rounding errors of p wide)
    if (!dense && threshold > 0.0 && threshold < 1.0 - 1e-4 && all_heads_write_objectness(net, in_dev, logits, batch))
        net->obj_min_logit = (float)(std::log(threshold / (1.0 - threshold)) - 0.01);
    rc = run_forward(net, in_dev, batch, logits, s);
    net->obj_min_logit = -INFINITY;
    if (rc) return rc;
    return run_decode_nms(net->head, logits, batch, threshold, iou_threshold, nms_mode, net->opt.cand_capacity,
                          net->opt.max_boxes, net->dev_ws + net->cand_off, reinterpret_cast<int *>(net->dev_ws + net->count_off),
                          boxes_dev, counts_dev, status_dev, nullptr, s, net->dev_ws + net->nms_off,
                          net->obj_valid ? reinterpret_cast<const float *>(net->dev_ws + net->obj_off) : nullptr, &net->cand_clean);
}

int yolo_net_read_layer(yolo_net *net, int layer, int batch, float *host_out, size_t n) {
    if (!net || !host_out) return fail(YOLO_ERR_ARG, "yolo_net_read_layer: null argument");
    if (!net->opt.keep_all) return fail(YOLO_ERR_STATE, "yolo_net_read_layer: create the net with keep_all=1");
    if (layer < 0 || layer >= (int)net->layers.size() || !net->layers[layer].materialised)
        return fail(YOLO_ERR_ARG, "yolo_net_read_layer: layer has no materialised tensor (fused away or out of range)");
    const View &v = net->layers[layer].view;
    if (v.buf < 0) return fail(YOLO_ERR_ARG, "yolo_net_read_layer: layer lives in a caller-owned tensor");
    const size_t need = (size_t)batch * v.H * v.W * v.C;
    if (n < need || batch > net->opt.max_batch) return fail(YOLO_ERR_ARG, "yolo_net_read_layer: host buffer too small");
    HIP_TRY(hipDeviceSynchronize());
    const Buffer &b = net->buffers[v.buf];
    const int es = v.f32 ? 4 : net->esize;
    std::vector<unsigned char> tmp((size_t)batch * v.img_stride * es);
    HIP_TRY(hipMemcpy(tmp.data(), net->dev_ws + b.offset, tmp.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < batch; ++i)
        for (long long px = 0; px < (long long)v.H * v.W; ++px)
            for (int c = 0; c < v.C; ++c) {
                const size_t e = (size_t)i * v.img_stride + (size_t)px * v.ld + v.coff + c;
                float f;
                if (es == 4) f = reinterpret_cast<const float *>(tmp.data())[e];
                else f = (float)reinterpret_cast<const _Float16 *>(tmp.data())[e];
                host_out[((size_t)i * v.H * v.W + px) * v.C + c] = f;
            }
    return YOLO_OK;
}

size_t yolo_decode_scratch_bytes(const yolo_head_desc *head, int batch, int cand_capacity) {
    (void)head;
    if (batch <= 0) return 0;
    if (cand_capacity <= 0) cand_capacity = 4096;
    return ((sizeof(Candidate) * (size_t)cand_capacity * batch + 255) & ~(size_t)255) + ((sizeof(int) * (size_t)batch * kCandCountStride + 255) & ~(size_t)255) +
           nms_scratch_bytes(cand_capacity) * (size_t)batch;
}

int yolo_decode_nms(const yolo_head_desc *head, const float *logits_dev, int batch, double threshold, double iou_threshold,
                    int nms_mode, int cand_capacity, int max_boxes, void *scratch_dev, size_t scratch_bytes, yolo_box *boxes_dev,
                    int32_t *counts_dev, int32_t *status_dev, void *stream) {
    if (!head || !logits_dev || !scratch_dev || !boxes_dev || !counts_dev || !status_dev || batch <= 0)
        return fail(YOLO_ERR_ARG, "yolo_decode_nms: null argument");
    if (cand_capacity <= 0) cand_capacity = 4096;
    if (max_boxes <= 0) max_boxes = 256;
    std::string err;
    int rc = check_head(head, 0, err);
    if (rc) return fail(rc, "yolo_decode_nms: " + err);
    if (cand_capacity > 65536) return fail(YOLO_ERR_ARG, "yolo_decode_nms: cand_capacity above 65536");
    if (scratch_bytes < yolo_decode_scratch_bytes(head, batch, cand_capacity)) return fail(YOLO_ERR_ARG, "yolo_decode_nms: scratch too small");
    unsigned char *cand = static_cast<unsigned char *>(scratch_dev);
    int *cnt = reinterpret_cast<int *>(cand + ((sizeof(Candidate) * (size_t)cand_capacity * batch + 255) & ~(size_t)255));
    unsigned char *slabs = reinterpret_cast<unsigned char *>(cnt) + ((sizeof(int) * (size_t)batch * kCandCountStride + 255) & ~(size_t)255);
    return run_decode_nms(*head, logits_dev, batch, threshold, iou_threshold, nms_mode, cand_capacity, max_boxes, cand, cnt,
                          boxes_dev, counts_dev, status_dev, nullptr, static_cast<hipStream_t>(stream), slabs);
}

int yolo_preprocess_resize(const uint8_t *src_dev, int src_h, int src_w, int src_row_bytes, float *dst_dev, int dst_h, int dst_w,
                           int swap_rb, void *stream) {
    if (!src_dev || !dst_dev || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || src_row_bytes < 3 * src_w)
        return fail(YOLO_ERR_ARG, "yolo_preprocess_resize: bad argument");
    ResizeParams p;
    p.src = src_dev; p.dst = dst_dev;
    p.src_h = src_h; p.src_w = src_w; p.src_row_bytes = src_row_bytes; p.dst_h = dst_h; p.dst_w = dst_w; p.swap_rb = swap_rb ? 1 : 0;
    HIP_TRY(launch_resize(p, static_cast<hipStream_t>(stream)));
    return YOLO_OK;
}

int yolo_nms_host(const double *xywh, const float *prob, const int32_t *class_idx, int n, double iou_threshold, int nms_mode,
                  int32_t *keep_idx, int32_t *n_keep) {
    if (n < 0 || !n_keep || (n && (!xywh || !prob || !class_idx || !keep_idx))) return fail(YOLO_ERR_ARG, "yolo_nms_host: null argument");
    *n_keep = 0;
    if (n == 0) return YOLO_OK;                      // base.py:196-197
    if (n > 65536) return fail(YOLO_ERR_OVERFLOW, "yolo_nms_host: more than 65536 boxes");
    std::vector<Candidate> c(n);
    for (int i = 0; i < n; ++i) {
        c[i].x = (float)xywh[4 * i]; c[i].y = (float)xywh[4 * i + 1]; c[i].w = xywh[4 * i + 2]; c[i].h = xywh[4 * i + 3];
        c[i].prob = prob[i]; c[i].cls = class_idx[i]; c[i].scan = (unsigned)i; c[i].pad_ = 0;
    }
    unsigned char *dev = nullptr;
    const size_t cb = sizeof(Candidate) * (size_t)n, bb = sizeof(yolo_box) * (size_t)n, ib = sizeof(int) * (size_t)n;
    const size_t total = ((cb + 255) & ~(size_t)255) + ((bb + 255) & ~(size_t)255) + ((ib + 255) & ~(size_t)255) + 768 + nms_scratch_bytes(n);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), total));
    unsigned char *d_c = dev, *d_b = d_c + ((cb + 255) & ~(size_t)255), *d_i = d_b + ((bb + 255) & ~(size_t)255);
    int *d_cnt = reinterpret_cast<int *>(d_i + ((ib + 255) & ~(size_t)255));    // [count, kept, status] 256 B apart
    int rc = YOLO_OK;
    do {
        if (hipMemcpy(d_c, c.data(), cb, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_cnt, &n, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(YOLO_ERR_HIP, "yolo_nms_host: H2D copy failed"); break; }
        NmsParams np;
        np.cand = reinterpret_cast<const Candidate *>(d_c);
        np.cand_count = d_cnt;
        np.reset_count = nullptr;
        np.cap = n; np.max_boxes = n; np.mode = nms_mode; np.iou_threshold = iou_threshold;
        np.boxes = reinterpret_cast<yolo_box *>(d_b);
        np.counts = d_cnt + 64; np.status = d_cnt + 128;
        np.keep_idx = reinterpret_cast<int *>(d_i);
        np.scratch = reinterpret_cast<unsigned char *>(d_cnt) + 768;      // only used above 4096 boxes
        np.scratch_stride = nms_scratch_bytes(n);
        hipError_t e = launch_nms(np, 1, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(YOLO_ERR_HIP, std::string("yolo_nms_host: ") + hipGetErrorString(e)); break; }
        int kept = 0;
        if (hipMemcpy(&kept, d_cnt + 64, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(keep_idx, d_i, sizeof(int) * (size_t)kept, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(YOLO_ERR_HIP, "yolo_nms_host: D2H copy failed"); break; }
        *n_keep = kept;
    } while (0);
    (void)hipFree(dev);
    return rc;
}

}  // extern "C"
