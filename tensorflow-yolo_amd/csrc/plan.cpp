// Planner: turns the reference-style layer list (net/layers.py vocabulary) into a short list of
// fused kernels over strided NHWC views.
//
//   * BatchNorm (layers.py:41-48) is folded into the conv weights/bias at load time.
//   * leaky ReLU (layers.py:50-51) and the bias add are the conv epilogue.
//   * shortcut (layers.py:100-103) becomes a residual read in the producing conv's epilogue.
//   * route/concat (layers.py:84-87) is zero-copy: producers write channel slices of one buffer.
//   * upsample (layers.py:112-116) / reorg (layers.py:90-97) become output index maps of the
//     producing conv.
//   * yolo_layer/detection_layer (layers.py:119-134) are views: the head convs write float32
//     straight into the reference-layout output tensor.
// Every fusion has a generic fallback kernel so any graph over the vocabulary still runs.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "yolo_internal.h"

namespace yolo {

static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
const char *get_error() { return g_err.c_str(); }

static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }
static inline size_t roundup_sz(size_t a, size_t b) { return (a + b - 1) / b * b; }

namespace {

struct Fuse {
    int kind = 0;       // 0 none, 1 residual, 2 upsample x2, 3 reorg x2
    int layer = -1;     // the fused shortcut / upsample / reorg layer
    int res = -1;       // residual source layer (resolved)
};

struct Planner {
    yolo_net *net;
    std::vector<LayerInfo> &L;
    int n;
    int esize, epc;
    std::vector<Fuse> fuse;               // per conv layer
    std::vector<std::vector<int>> effc;   // effective consumers (through single-source routes)
    std::vector<View> claim;              // per layer: placement inside a concat buffer
    std::vector<char> has_claim;
    std::vector<View> head_target;        // per conv layer: direct write into the user output
    std::vector<char> has_head;
    struct Copy { int route, src, coff; };
    std::vector<Copy> copies;
    std::vector<int> route_buf;
    bool first_direct = false;
    std::string err;

    Planner(yolo_net *n_) : net(n_), L(n_->layers), n((int)n_->layers.size()) {}

    int resolve(int i) const {
        while (L[i].d.op == YOLO_OP_ROUTE && L[i].d.n_src == 1) i = L[i].d.src[0];
        return i;
    }
    bool fail(const std::string &s) { err = s; return false; }

    int new_buffer(long long elems_per_image, int es, bool concat = false) {
        Buffer b;
        b.elems_per_image = elems_per_image;
        b.esize = es;
        b.is_concat = concat;
        net->buffers.push_back(b);
        return (int)net->buffers.size() - 1;
    }
    View dense_view(int buf, int H, int W, int C, int ld) {
        View v;
        v.buf = buf; v.H = H; v.W = W; v.C = C; v.ld = ld; v.coff = 0;
        v.img_stride = (long long)H * W * ld;
        return v;
    }
    View alloc_view(int H, int W, int C) {
        return dense_view(new_buffer((long long)H * W * C, esize), H, W, C, C);
    }

    bool shapes() {
        for (int i = 0; i < n; ++i) {
            yolo_layer_desc &d = L[i].d;
            char nm[64];
            snprintf(nm, sizeof nm, "layer %d: ", i);
            if (d.n_src < 0 || d.n_src > YOLO_MAX_SRC) return fail(std::string(nm) + "bad n_src");
            for (int k = 0; k < d.n_src; ++k)
                if (d.src[k] < 0 || d.src[k] >= i) return fail(std::string(nm) + "source index must precede the layer");
            auto S = [&](int k) -> LayerInfo & { return L[d.src[k]]; };
            switch (d.op) {
            case YOLO_OP_INPUT:
                if (i != 0) return fail(std::string(nm) + "input layer must be layer 0");
                if (d.h <= 0 || d.w <= 0 || d.c <= 0) return fail(std::string(nm) + "bad input shape");
                L[i].H = d.h; L[i].W = d.w; L[i].C = d.c;
                break;
            case YOLO_OP_CONV:
                if (d.n_src != 1) return fail(std::string(nm) + "conv takes one source");
                if (d.ksize != 1 && d.ksize != 3) return fail(std::string(nm) + "conv ksize must be 1 or 3");
                if (d.stride != 1 && d.stride != 2) return fail(std::string(nm) + "conv stride must be 1 or 2");
                if (d.filters <= 0) return fail(std::string(nm) + "conv filters must be positive");
                // stride 1 -> SAME; stride 2 -> pad (k-1) then VALID (layers.py:28-30)
                L[i].H = d.stride == 1 ? S(0).H : (S(0).H - 1) / d.stride + 1;
                L[i].W = d.stride == 1 ? S(0).W : (S(0).W - 1) / d.stride + 1;
                L[i].C = d.filters;
                break;
            case YOLO_OP_MAXPOOL:
                if (d.n_src != 1 || d.ksize != 2 || (d.stride != 1 && d.stride != 2))
                    return fail(std::string(nm) + "maxpool supports ksize 2, stride 1|2");
                L[i].H = d.stride == 1 ? S(0).H : (S(0).H - 1) / 2 + 1;
                L[i].W = d.stride == 1 ? S(0).W : (S(0).W - 1) / 2 + 1;
                L[i].C = S(0).C;
                break;
            case YOLO_OP_ROUTE: {
                if (d.n_src < 1) return fail(std::string(nm) + "route needs a source");
                int c = 0;
                for (int k = 0; k < d.n_src; ++k) {
                    if (S(k).H != S(0).H || S(k).W != S(0).W) return fail(std::string(nm) + "route sources differ in H/W");
                    c += S(k).C;
                }
                L[i].H = S(0).H; L[i].W = S(0).W; L[i].C = c;
                break;
            }
            case YOLO_OP_REORG:
                if (d.n_src != 1 || d.stride != 2 || (S(0).H & 1) || (S(0).W & 1))
                    return fail(std::string(nm) + "reorg supports stride 2 on even H/W");
                L[i].H = S(0).H / 2; L[i].W = S(0).W / 2; L[i].C = S(0).C * 4;
                break;
            case YOLO_OP_SHORTCUT:
                if (d.n_src != 2 || S(0).H != S(1).H || S(0).W != S(1).W || S(0).C != S(1).C)
                    return fail(std::string(nm) + "shortcut needs two sources of equal shape");
                L[i].H = S(0).H; L[i].W = S(0).W; L[i].C = S(0).C;
                break;
            case YOLO_OP_UPSAMPLE:
                if (d.n_src != 1 || d.stride != 2) return fail(std::string(nm) + "upsample supports stride 2");
                L[i].H = S(0).H * 2; L[i].W = S(0).W * 2; L[i].C = S(0).C;
                break;
            case YOLO_OP_YOLO:
                if (d.n_src != 1 || d.n_anchors < 1 || d.n_anchors > YOLO_MAX_ANCHORS || S(0).C % d.n_anchors)
                    return fail(std::string(nm) + "yolo layer: channels must divide by the anchor count");
                L[i].H = S(0).H; L[i].W = S(0).W; L[i].C = S(0).C;
                break;
            case YOLO_OP_DETECTION:
                if (i != n - 1) return fail(std::string(nm) + "detection layer must be last");
                if (d.n_src < 1 || d.n_src > YOLO_MAX_SCALES) return fail(std::string(nm) + "detection takes 1..4 yolo layers");
                for (int k = 0; k < d.n_src; ++k)
                    if (S(k).d.op != YOLO_OP_YOLO) return fail(std::string(nm) + "detection sources must be yolo layers");
                break;
            default:
                return fail(std::string(nm) + "unknown op");
            }
        }
        if (L[0].d.op != YOLO_OP_INPUT) return fail("layer 0 must be the input layer");
        return true;
    }

    void consumers() {
        effc.assign(n, {});
        for (int i = 1; i < n; ++i) {
            const yolo_layer_desc &d = L[i].d;
            for (int k = 0; k < d.n_src; ++k) L[d.src[k]].consumers.push_back(i);
            if (d.op == YOLO_OP_ROUTE && d.n_src == 1) continue;   // an alias does not consume
            for (int k = 0; k < d.n_src; ++k) effc[resolve(d.src[k])].push_back(i);
        }
    }

    bool sole(int a, int i) const { return effc[a].size() == 1 && effc[a][0] == i; }

    void fusions() {
        fuse.assign(n, Fuse());
        for (int i = 1; i < n; ++i) {
            const yolo_layer_desc &d = L[i].d;
            if (d.op == YOLO_OP_SHORTCUT) {
                int a = resolve(d.src[0]), b = resolve(d.src[1]);
                for (int t = 0; t < 2; ++t) {
                    if (a != b && L[a].d.op == YOLO_OP_CONV && sole(a, i) && fuse[a].kind == 0) {
                        fuse[a].kind = 1; fuse[a].layer = i; fuse[a].res = b;
                        L[i].fused_into = a;
                        break;
                    }
                    std::swap(a, b);
                }
            } else if (d.op == YOLO_OP_UPSAMPLE || d.op == YOLO_OP_REORG) {
                int a = resolve(d.src[0]);
                if (L[a].d.op == YOLO_OP_CONV && sole(a, i) && fuse[a].kind == 0) {
                    fuse[a].kind = d.op == YOLO_OP_UPSAMPLE ? 2 : 3;
                    fuse[a].layer = i;
                    L[i].fused_into = a;
                }
            }
        }
    }

    // The layer index whose logical output a source refers to (after single-source routes).
    bool claimable(int r) const {
        int op = L[r].d.op;
        if (op == YOLO_OP_INPUT || op == YOLO_OP_YOLO || op == YOLO_OP_DETECTION) return false;
        if (op == YOLO_OP_ROUTE) return false;               // nested concat: copy instead
        if (op == YOLO_OP_CONV && fuse[r].kind != 0) return false;   // its raw output does not exist
        if (r == n - 1 || has_claim[r] || has_head[r]) return false;
        return true;
    }

    void heads() {
        head_target.assign(n, View());
        has_head.assign(n, 0);
        yolo_head_desc &hd = net->head;
        memset(&hd, 0, sizeof hd);
        const yolo_layer_desc &last = L[n - 1].d;
        if (last.op != YOLO_OP_DETECTION) return;
        long long total = 0;
        for (int k = 0; k < last.n_src; ++k) total += (long long)L[last.src[k]].H * L[last.src[k]].W * L[last.src[k]].C;
        net->out_count = (size_t)total;
        long long off = 0;
        hd.version = 3;
        hd.n_scales = last.n_src;
        for (int k = 0; k < last.n_src; ++k) {
            int y = last.src[k];
            const yolo_layer_desc &yd = L[y].d;
            hd.h[k] = L[y].H; hd.w[k] = L[y].W; hd.n_anchors[k] = yd.n_anchors;
            hd.n_classes = L[y].C / yd.n_anchors - 5;
            for (int a = 0; a < 2 * yd.n_anchors; ++a) hd.anchors[k][a] = yd.anchors[a];
            int a = resolve(yd.src[0]);
            View v;
            v.buf = BUF_USER_OUT; v.H = L[y].H; v.W = L[y].W; v.C = L[y].C; v.ld = L[y].C; v.coff = 0;
            v.img_stride = total; v.base = off; v.f32 = true;
            if (L[a].d.op == YOLO_OP_CONV && sole(a, y) && fuse[a].kind == 0 && !has_head[a]) {
                head_target[a] = v;
                has_head[a] = 1;
            }
            L[y].view = v;      // where the yolo layer's rows live in the output
            off += (long long)L[y].H * L[y].W * L[y].C;
        }
    }

    void claims() {
        claim.assign(n, View());
        has_claim.assign(n, 0);
        route_buf.assign(n, BUF_NONE);
        for (int i = 1; i < n; ++i) {
            const yolo_layer_desc &d = L[i].d;
            if (d.op != YOLO_OP_ROUTE || d.n_src < 2) continue;
            int buf = new_buffer((long long)L[i].H * L[i].W * L[i].C, esize, true);
            route_buf[i] = buf;
            int off = 0;
            for (int k = 0; k < d.n_src; ++k) {
                int r = resolve(d.src[k]);
                if (claimable(r)) {
                    View v = dense_view(buf, L[r].H, L[r].W, L[r].C, L[i].C);
                    v.coff = off;
                    claim[r] = v;
                    has_claim[r] = 1;
                } else {
                    copies.push_back({i, r, off});
                }
                off += L[r].C;
            }
        }
    }

    View out_view_for(int key) {
        if (key == n - 1) {     // final layer: dense float32 in the user's output tensor
            View v = dense_view(BUF_USER_OUT, L[key].H, L[key].W, L[key].C, L[key].C);
            v.f32 = true;
            net->out_count = (size_t)L[key].H * L[key].W * L[key].C;
            return v;
        }
        if (has_claim[key]) return claim[key];
        return alloc_view(L[key].H, L[key].W, L[key].C);
    }

    void add_eltwise(int layer, const View &a, const View *b, const View &out, int outmode, const char *note) {
        Kernel k;
        k.kind = K_ELTWISE; k.layer = layer; k.in = a;
        if (b) { k.in2 = *b; k.has_res = 1; }
        k.out = out; k.outmode = outmode; k.note = note;
        net->kernels.push_back(k);
    }

    bool emit() {
        size_t wsrc = 0;
        size_t woff = 0;
        double flops = 0;
        for (int i = 0; i < n; ++i) {
            const yolo_layer_desc &d = L[i].d;
            char nm[64];
            snprintf(nm, sizeof nm, "layer %d: ", i);
            switch (d.op) {
            case YOLO_OP_INPUT: {
                // first layer 3x3/1 on a 3-channel input with 16|32 filters: the direct kernel of
                // first.hip reads the caller's float32 tensor itself, so no cast/pad pass is needed
                first_direct = n > 2 && d.c == 3 && L[1].d.op == YOLO_OP_CONV && L[1].d.src[0] == 0 && L[1].d.ksize == 3 &&
                               L[1].d.stride == 1 && (L[1].d.filters == 16 || L[1].d.filters == 32) && sole(0, 1) &&
                               fuse[1].kind == 0 && !has_head[1] && !has_claim[1] && !getenv("YOLO_NO_FIRST_DIRECT");
                if (first_direct) {
                    L[i].view = dense_view(BUF_USER_IN, d.h, d.w, d.c, d.c);
                    L[i].view.f32 = true;
                    L[i].materialised = true;
                    break;
                }
                int cpad = roundup(d.c, epc);
                Kernel k;
                k.kind = K_PREP; k.layer = 0;
                k.in = dense_view(BUF_USER_IN, d.h, d.w, d.c, d.c);
                k.in.f32 = true;
                View v = dense_view(new_buffer((long long)d.h * d.w * cpad, esize), d.h, d.w, d.c, cpad);
                k.out = v;
                k.note = "f32 NHWC -> T NHWC, channels zero-padded to a 16-byte chunk";
                net->kernels.push_back(k);
                L[i].view = v; L[i].materialised = true;
                break;
            }
            case YOLO_OP_CONV: {
                int s = resolve(d.src[0]);
                if (!L[s].materialised) return fail(std::string(nm) + "source not materialised");
                if (i == 1 && first_direct) {
                    Kernel k;
                    k.kind = K_FIRST; k.layer = 1; k.src_layer = 1;
                    k.in = L[0].view;
                    k.ksize = 3; k.stride = 1; k.cout = d.filters; k.cin = 3; k.cin_s = 3;
                    k.leaky = d.leaky; k.batch_norm = d.batch_norm;
                    k.out = out_view_for(1);
                    if (has_claim[1]) k.note = "-> concat slice";
                    k.note += " direct conv on the float32 input (no cast/pad pass)";
                    k.w_src = wsrc;
                    wsrc += (size_t)d.filters * 27 + (d.batch_norm ? 4 : 1) * (size_t)d.filters;
                    k.w_off = woff; k.w_bytes = (size_t)27 * d.filters * 4;
                    k.b_off = roundup_sz(k.w_off + k.w_bytes, 256);
                    woff = roundup_sz(k.b_off + (size_t)d.filters * 4, 256);
                    flops += 2.0 * L[i].H * L[i].W * d.filters * 27;
                    net->kernels.push_back(k);
                    L[1].view = k.out; L[1].materialised = true;
                    break;
                }
                const View &in = L[s].view;
                if (in.f32) return fail(std::string(nm) + "conv cannot read a float32 head tensor");
                int cin = L[s].C;
                int cin_s = roundup(cin, epc);
                if (cin_s > in.ld - in.coff && cin_s != cin) return fail(std::string(nm) + "padded input channels exceed the view");
                if (cin_s != cin && L[s].d.op != YOLO_OP_INPUT)
                    return fail(std::string(nm) + "input channels must be a multiple of the 16-byte chunk");
                int chunks = cin_s / epc;
                Kernel k;
                k.kind = K_CONV; k.src_layer = i;
                k.in = in;
                k.ksize = d.ksize; k.stride = d.stride; k.cout = d.filters; k.cin = cin; k.cin_s = cin_s;
                k.leaky = d.leaky; k.batch_norm = d.batch_norm;
                int taps = d.ksize * d.ksize;
                if (chunks % 8 == 0) {
                    k.perchunk = 0; k.cpt = chunks; k.ktiles = taps * (chunks / 8);
                } else if (chunks == 1 || chunks == 2 || chunks == 4) {
                    k.perchunk = 1; k.cpt = chunks; k.ktiles = (taps * chunks + 7) / 8;
                } else {
                    return fail(std::string(nm) + "unsupported input channel count for the implicit-GEMM tiling");
                }
                k.cfg = d.filters <= 32 ? CFG_N32 : d.filters <= 64 ? CFG_N64 : CFG_N128;
                int key = i;
                const Fuse &f = fuse[i];
                if (f.kind) key = f.layer;
                if (f.kind == 1) {
                    if (!L[f.res].materialised) return fail(std::string(nm) + "residual source not materialised");
                    k.in2 = L[f.res].view; k.has_res = 1;
                    if (k.in2.f32) return fail(std::string(nm) + "residual cannot be a float32 head tensor");
                    k.note = "fused: +shortcut(layer " + std::to_string(f.layer) + ")";
                } else if (f.kind == 2) {
                    k.outmode = OUT_UP2; k.note = "fused: upsample x2 (layer " + std::to_string(f.layer) + ")";
                } else if (f.kind == 3) {
                    k.outmode = OUT_REORG2; k.note = "fused: reorg x2 (layer " + std::to_string(f.layer) + ")";
                }
                k.layer = key;
                if (has_head[i]) {
                    k.head = 1;
                    k.out = head_target[i];
                    k.note += " -> head logits (float32, reference layout)";
                } else {
                    k.out = out_view_for(key);
                }
                if (has_claim[key]) k.note += " -> concat slice";
                // weights
                int cout_pad = roundup(d.filters, 128);
                size_t wrow = (size_t)k.ktiles * 128;
                k.w_src = wsrc;
                wsrc += (size_t)d.filters * cin * taps + (d.batch_norm ? 4 : 1) * (size_t)d.filters;
                k.w_off = woff; k.w_bytes = wrow * cout_pad;
                k.b_off = roundup_sz(k.w_off + k.w_bytes, 256);
                // (bias region padded to a multiple of 256 couts: the 256-cout tiles load the bias of their whole tile as the
                // accumulators' initial value before they know how many of its couts exist -- conv_common.h: conv_init_acc_bias)
                woff = roundup_sz(k.b_off + (size_t)roundup(d.filters, 256) * 4, 256);
                flops += 2.0 * L[i].H * L[i].W * d.filters * taps * cin;
                // Darknet-53 stem: first-layer kernel + this 3x3/2 32->64 conv as one kernel (stem.hip) when nobody else
                // reads the 32-channel tensor (keep_all needs it in memory) and the output takes 16-byte stores
                if (i == 2 && first_direct && !net->kernels.empty() && net->kernels.back().kind == K_FIRST && s == 1 && sole(1, 2) &&
                    !net->opt.keep_all && net->opt.dtype == YOLO_DTYPE_F16 && d.ksize == 3 && d.stride == 2 && d.filters == 64 &&
                    L[1].d.filters == 32 && d.leaky && L[1].d.leaky && f.kind == 0 && !has_head[i] && L[1].H % 2 == 0 && L[1].W % 2 == 0 &&
                    k.out.ld % epc == 0 && (k.out.base + k.out.coff) % epc == 0 && k.out.img_stride % epc == 0 && !k.out.f32 &&
                    !getenv("YOLO_NO_STEM")) {
                    net->kernels.back().stem = 1;
                    k.stem = 2;
                    k.note += " fused with the first layer (stem.hip): the 32-channel tensor stays in LDS";
                }
                // ... and the 1x1 64->32 conv behind the stem (Darknet-53 layer 3) is computed by the stem kernel too
                if (i == 3 && net->kernels.size() >= 2 && net->kernels.back().stem == 2 && s == 2 && d.ksize == 1 && d.stride == 1 &&
                    d.filters == 32 && cin == 64 && d.leaky && f.kind == 0 && !has_head[i] && !k.out.f32 && k.out.ld % epc == 0 &&
                    (k.out.base + k.out.coff) % epc == 0 && k.out.img_stride % epc == 0 && k.in.ld == net->kernels.back().out.ld &&
                    k.in.coff == net->kernels.back().out.coff && k.in.buf == net->kernels.back().out.buf && !getenv("YOLO_NO_STEM3")) {
                    k.stem = 3;
                    k.note += " computed inside the stem kernel (no launch)";
                }
                // Back-to-back 1x1 (conv_common.h: conv_epilogue_fused_1x1): a 1x1 128 -> 64 conv whose input is exactly what the 3x3
                // conv in front of it writes, where that conv's workgroups hold all 128 couts of their positions (Darknet-53 at
                // 152 x 152: the stride-2 conv into the stage and the first residual block's 3x3, each followed by the next block's
                // 1x1).  Structural conditions here; whether a launch takes the fused instantiation depends on the tile its batch
                // picks (api.cpp: conv_fuse2), else the 1x1 runs as a launch of its own.
                if (!net->kernels.empty() && net->opt.dtype == YOLO_DTYPE_F16 && !net->opt.keep_all && !getenv("YOLO_NO_FUSE2") &&
                    d.ksize == 1 && d.stride == 1 && d.filters == 64 && cin == 128 && f.kind == 0 && !has_head[i] && !k.out.f32 &&
                    k.out.ld % epc == 0 && (k.out.base + k.out.coff) % epc == 0 && k.out.img_stride % epc == 0) {
                    Kernel &c = net->kernels.back();
                    if (c.kind == K_CONV && c.layer == s && c.stem == 0 && c.cout == 128 && c.ksize == 3 && c.cpt % 4 == 0 &&
                        c.outmode == OUT_NORMAL && !c.head && !c.out.f32 && c.out.buf == k.in.buf && c.out.ld == k.in.ld &&
                        c.out.coff == k.in.coff && c.out.base == k.in.base && c.out.img_stride == k.in.img_stride) {
                        c.fuse2_next = 1;
                        k.fuse2_prev = 1;
                        k.note += " (computed by the conv in front of it where that launch holds all 128 channels per workgroup)";
                    }
                }
                net->kernels.push_back(k);
                L[key].view = k.out; L[key].materialised = true;
                if (key != i) { L[i].materialised = false; }
                break;
            }
            case YOLO_OP_MAXPOOL: {
                int s = resolve(d.src[0]);
                if (!L[s].materialised) return fail(std::string(nm) + "source not materialised");
                // Darknet-19 / tiny-YOLO: the 2x2/2 pool right behind the first conv is taken inside the first-layer kernel
                // (the full-resolution tensor is never written) when nobody else reads that tensor
                if (i == 2 && s == 1 && first_direct && !net->kernels.empty() && net->kernels.back().kind == K_FIRST && sole(1, 2) &&
                    !net->opt.keep_all && d.stride == 2 && L[1].H % 2 == 0 && L[1].W % 2 == 0 && !getenv("YOLO_NO_FIRST_POOL")) {
                    View pv = out_view_for(i);
                    if (!pv.f32 && pv.ld % epc == 0 && (pv.base + pv.coff) % epc == 0 && pv.img_stride % epc == 0) {
                        Kernel &f = net->kernels.back();
                        f.pool_fused = 1;
                        f.out = pv;
                        f.layer = i;
                        f.note += " + fused 2x2/2 max-pool (layer 2)";
                        L[i].view = pv; L[i].materialised = true;
                        L[1].materialised = false;
                        break;
                    }
                }
                // conv 3x3/1 -> max-pool 2x2/2 on a wide map: the pool is taken in the conv's epilogue (conv_common.h:
                // conv_epilogue_pool2, 2-D tap tiles only, so the conv is pinned to one: 13 = 64 couts, 12 = 128-cout tiles) and the
                // full-resolution tensor is never written.  Wide maps only: below ~96 columns the padded-linear tiles beat the 2-D
                // ones by more than the pool kernel costs (r03 sweep: 52 x 52 128 -> 256 +9 us vs a 9 us pool)
                if (!net->kernels.empty() && net->kernels.back().kind == K_CONV && net->kernels.back().layer == s &&
                    net->kernels.back().src_layer == s && sole(s, i) && !net->opt.keep_all && d.stride == 2 && L[s].H % 2 == 0 &&
                    L[s].W % 2 == 0 && L[s].W >= 96 && !getenv("YOLO_NO_CONV_POOL")) {
                    Kernel &c = net->kernels.back();
                    const bool f16 = net->opt.dtype == YOLO_DTYPE_F16;
                    const bool tile13 = c.cout == 64, tile17 = c.cout == 32, tile12 = c.cout > 64 && f16;       // (the float32 2-D tiles: 64 and 32 couts)
                    if (c.ksize == 3 && c.stride == 1 && c.outmode == OUT_NORMAL && !c.has_res && !c.head && !c.stem && !has_claim[s] &&
                        c.cpt % 4 == 0 && c.cout % 16 == 0 && (tile13 || tile12 || tile17)) {
                        View pv = out_view_for(i);
                        if (!pv.f32 && pv.ld % epc == 0 && (pv.base + pv.coff) % epc == 0 && pv.img_stride % epc == 0) {
                            c.outmode = OUT_POOL2;
                            c.out = pv;
                            c.layer = i;
                            c.tile = tile13 ? 13 : tile17 ? 17 : 12;
                            c.note += " + fused 2x2/2 max-pool (layer " + std::to_string(i) + ")";
                            if (has_claim[i]) c.note += " -> concat slice";
                            L[i].view = pv; L[i].materialised = true;
                            L[s].materialised = false;
                            break;
                        }
                    }
                }
                Kernel k;
                k.kind = K_POOL; k.layer = i; k.in = L[s].view; k.pool_stride = d.stride;
                if (k.in.f32) return fail(std::string(nm) + "maxpool cannot read a float32 head tensor");
                k.out = out_view_for(i);
                if (k.out.f32) {    // final layer: pool into T, then convert
                    View t = alloc_view(L[i].H, L[i].W, L[i].C);
                    View fin = k.out;
                    k.out = t;
                    net->kernels.push_back(k);
                    add_eltwise(i, t, nullptr, fin, OUT_NORMAL, "convert final layer to float32");
                    L[i].view = fin;
                } else {
                    net->kernels.push_back(k);
                    L[i].view = k.out;
                }
                L[i].materialised = true;
                break;
            }
            case YOLO_OP_ROUTE: {
                if (d.n_src == 1) {
                    int s = resolve(d.src[0]);
                    L[i].view = L[s].view; L[i].materialised = L[s].materialised;
                    if (i == n - 1) {
                        View fin = out_view_for(i);
                        add_eltwise(i, L[s].view, nullptr, fin, OUT_NORMAL, "convert final layer to float32");
                        L[i].view = fin;
                    }
                    break;
                }
                View v = dense_view(route_buf[i], L[i].H, L[i].W, L[i].C, L[i].C);
                for (const Copy &c : copies) {
                    if (c.route != i) continue;
                    if (!L[c.src].materialised) return fail(std::string(nm) + "route source not materialised");
                    View o = dense_view(route_buf[i], L[c.src].H, L[c.src].W, L[c.src].C, L[i].C);
                    o.coff = c.coff;
                    add_eltwise(i, L[c.src].view, nullptr, o, OUT_NORMAL, "concat by copy (source could not write in place)");
                }
                L[i].view = v; L[i].materialised = true;
                if (i == n - 1) {
                    View fin = out_view_for(i);
                    add_eltwise(i, v, nullptr, fin, OUT_NORMAL, "convert final layer to float32");
                    L[i].view = fin;
                }
                break;
            }
            case YOLO_OP_REORG:
            case YOLO_OP_UPSAMPLE:
            case YOLO_OP_SHORTCUT: {
                if (L[i].fused_into >= 0) break;    // produced by the conv's epilogue
                int a = resolve(d.src[0]);
                if (!L[a].materialised) return fail(std::string(nm) + "source not materialised");
                View out = out_view_for(i);
                if (d.op == YOLO_OP_SHORTCUT) {
                    int b = resolve(d.src[1]);
                    if (!L[b].materialised) return fail(std::string(nm) + "source not materialised");
                    add_eltwise(i, L[a].view, &L[b].view, out, OUT_NORMAL, "standalone shortcut add");
                } else {
                    add_eltwise(i, L[a].view, nullptr, out, d.op == YOLO_OP_REORG ? OUT_REORG2 : OUT_UP2,
                                d.op == YOLO_OP_REORG ? "standalone reorg" : "standalone upsample");
                }
                L[i].view = out; L[i].materialised = true;
                break;
            }
            case YOLO_OP_YOLO: {
                int a = resolve(d.src[0]);
                if (n - 1 > i && L[n - 1].d.op == YOLO_OP_DETECTION) {
                    bool in_det = false;
                    for (int k = 0; k < L[n - 1].d.n_src; ++k) in_det |= L[n - 1].d.src[k] == i;
                    if (in_det) {
                        if (!(L[a].d.op == YOLO_OP_CONV && has_head[a])) {
                            if (!L[a].materialised) return fail(std::string(nm) + "source not materialised");
                            add_eltwise(i, L[a].view, nullptr, L[i].view, OUT_NORMAL, "copy head rows into the output tensor");
                        }
                        L[i].materialised = true;
                        break;
                    }
                }
                L[i].view = L[a].view; L[i].materialised = L[a].materialised;
                break;
            }
            case YOLO_OP_DETECTION:
                L[i].materialised = true;
                break;
            }
        }
        // a final layer produced by something that cannot write the user tensor directly
        {
            int last = n - 1;
            int op = L[last].d.op;
            bool fused_last = (op == YOLO_OP_SHORTCUT || op == YOLO_OP_UPSAMPLE || op == YOLO_OP_REORG);
            if (op == YOLO_OP_INPUT) {
                View fin = out_view_for(last);
                add_eltwise(last, L[last].view, nullptr, fin, OUT_NORMAL, "convert final layer to float32");
                L[last].view = fin;
            }
            (void)fused_last;
            if (!L[last].materialised) return fail("final layer was not materialised");
        }
        net->weight_count = wsrc;
        net->weights_bytes = woff ? woff : 256;
        net->flops_per_image = flops;
        return true;
    }

    // BRANCH TAILS.  YOLOv3's heads end side branches: the 3x3 512 -> 1024 + 1x1 -> 255 behind the fifth conv of the 19 x 19 branch read
    // that conv's output and are read by nothing but the decode, while the list goes on with the route -> 1x1 -> upsample into the
    // 38 x 38 branch (net/v3.py:60-75).  Such a run of kernels -- a head conv that is not the last kernel plus the convs in front of it
    // whose output has that one reader -- may run on a second stream beside the kernels that follow it: the short latency-bound
    // launches behind the fork (1x1 + upsample, 1x1 on the concat) fill the CUs the one-workgroup-per-CU 19 x 19 tile leaves half
    // empty.  The buffers a tail touches stay alive to the end of the pass (allocate()): nothing that runs beside it is given their bytes.
    void side_chains() {
        std::vector<Kernel> &K = net->kernels;
        const int n = (int)K.size();
        auto readers = [&](int buf) { int c = 0; for (const Kernel &k : K) c += (k.in.buf == buf) + (k.has_res && k.in2.buf == buf); return c; };
        auto writers = [&](int buf) { int c = 0; for (const Kernel &k : K) c += k.out.buf == buf; return c; };
        int id = 0;
        for (int h = 0; h + 1 < n && id < 4; ++h) {
            if (K[h].kind != K_CONV || !K[h].head || K[h].side) continue;
            int a = h;
            while (a - 1 >= 0) {
                const Kernel &c = K[a], &pr = K[a - 1];
                if (pr.kind != K_CONV || pr.head || pr.stem || pr.side || pr.fuse2_next || pr.fuse2_prev || c.has_res) break;
                if (c.in.buf < 0 || pr.out.buf != c.in.buf || readers(c.in.buf) != 1 || writers(c.in.buf) != 1) break;
                --a;
            }
            if (h - a + 1 < 2) continue;        // (a head alone: nothing to hide behind)
            ++id;
            for (int k = a; k <= h; ++k) {
                K[k].side = id;
                K[k].note += " [branch tail " + std::to_string(id) + ": may run on a second stream beside the kernels behind it]";
            }
        }
        net->side_chains = id;
    }

    void allocate() {
        std::vector<Buffer> &B = net->buffers;
        std::vector<Kernel> &K = net->kernels;
        for (int k = 0; k < (int)K.size(); ++k) {
            for (const View *v : {&K[k].in, &K[k].in2, &K[k].out}) {
                if (v->buf < 0) continue;
                B[v->buf].first = std::min(B[v->buf].first, k);
                B[v->buf].last = std::max(B[v->buf].last, K[k].side ? (int)K.size() - 1 : k);      // (a branch tail's tensors: to the end of the pass)
            }
            // back-to-back 1x1: its output is written by the launch of the conv IN FRONT of it -- alive one kernel earlier, or it
            // could be given the bytes of a tensor that launch still reads (its own input dies there)
            if (K[k].fuse2_prev && k > 0 && K[k].out.buf >= 0) B[K[k].out.buf].first = std::min(B[K[k].out.buf].first, k - 1);
        }
        const size_t align = 4096;
        // Multi-stream forward (yolo_net_options.streams / YOLO_STREAMS): the batch runs as independent parts, each in its
        // own arena planned for its share of the batch -- lifetime-based reuse packs tensors of different per-image size into the same
        // bytes, so two halves at different layers must not share an arena.  Same total memory.
        // streams = 0 is the library's own rule, from interleaved one-stream / two-stream runs on one box (profiles/r04_streams_ab.jsonl):
        // two parts win where a half batch still fills the chip at every layer AND the net is a long chain of short launches
        // (YOLOv3-608 b32 +4.5 %, b16 +4.6 %, YOLOv3-416 b32 +5.8 %), lose where the halves get small (YOLOv3-416 b16 -4.5 %,
        // YOLOv3-608 b8 -5 %, YOLOv2-416 b16 -18 %) and do nothing for the short Darknet-19 chains (YOLOv2-416 b32 / b64 +-0.3 %,
        // tiny-YOLOv2 b64 float32 -0.5 %): fp16, >= 40 conv launches, >= 2.5 M input pixels per part.
        {
            int want = net->opt.streams > 0 ? net->opt.streams : (getenv("YOLO_STREAMS") ? atoi(getenv("YOLO_STREAMS")) : 0);
            const bool by_rule = want <= 0;
            if (want <= 0) {
                int convs = 0;
                for (const Kernel &k : K) convs += k.kind == K_CONV && k.stem < 3;
                const yolo_layer_desc &d0 = net->layers[0].d;
                const double px_per_part = (double)(net->opt.max_batch / 2) * d0.h * d0.w;
                want = (net->opt.dtype == YOLO_DTYPE_F16 && convs >= 40 && px_per_part >= 2.5e6) ? 2 : 1;
            }
            net->arenas = (want >= 2 && !net->opt.keep_all) ? (want > 4 ? 4 : want) : 1;
            if (net->arenas > net->opt.max_batch) net->arenas = net->opt.max_batch;
            // by rule: both arenas hold a full batch (twice the activation memory: 3 GB for YOLOv3-608 b32), so that the choice between
            // one pass and two halves can be made -- and changed -- per device (yolo_net_tune_streams)
            net->arena_full = by_rule && net->arenas == 2;
            net->parts = net->arenas;
        }
        const int arena_batch = net->arena_full ? net->opt.max_batch : (net->opt.max_batch + net->arenas - 1) / net->arenas;
        const size_t guard = net->opt.guard_bytes > 0 ? (size_t)net->opt.guard_bytes : 0;      // (test hook: yolo_net_options.guard_bytes)
        for (Buffer &b : B) {
            b.used = (size_t)b.elems_per_image * b.esize * arena_batch;
            b.bytes = roundup_sz(b.used + 256 + guard, align);
        }
        size_t top = 0;
        if (net->opt.keep_all) {
            for (Buffer &b : B) { b.offset = top; top += b.bytes; }
        } else {
            std::vector<int> order(B.size());
            for (size_t i = 0; i < B.size(); ++i) order[i] = (int)i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return B[a].first < B[b].first; });
            std::vector<int> placed;
            for (int id : order) {
                Buffer &b = B[id];
                if (b.last < 0) { b.offset = 0; b.bytes = 0; continue; }     // never used
                // gather live intervals that overlap in time, sorted by offset; first fit
                std::vector<std::pair<size_t, size_t>> busy;
                for (int p : placed)
                    if (!(B[p].last < b.first || B[p].first > b.last)) busy.push_back({B[p].offset, B[p].offset + B[p].bytes});
                std::sort(busy.begin(), busy.end());
                size_t at = 0;
                for (auto &iv : busy) {
                    if (at + b.bytes <= iv.first) break;
                    at = std::max(at, iv.second);
                }
                b.offset = at;
                top = std::max(top, at + b.bytes);
                placed.push_back(id);
            }
        }
        net->arena_bytes = roundup_sz(top, align);
        net->act_bytes = net->arena_bytes * net->arenas;
    }
};

}  // namespace

int plan_network(yolo_net *net, const yolo_layer_desc *layers, int n, std::string &err) {
    if (n < 2) { err = "need at least an input layer and one more layer"; return YOLO_ERR_ARG; }
    net->esize = net->opt.dtype == YOLO_DTYPE_F16 ? 2 : 4;
    net->epc = 16 / net->esize;
    net->layers.resize(n);
    for (int i = 0; i < n; ++i) net->layers[i].d = layers[i];
    Planner P(net);
    P.esize = net->esize; P.epc = net->epc;
    if (!P.shapes()) { err = P.err; return YOLO_ERR_PLAN; }
    P.consumers();
    P.fusions();
    P.heads();
    P.claims();
    if (!P.emit()) { err = P.err; return YOLO_ERR_PLAN; }
    P.side_chains();
    P.allocate();
    // workspace tail: head logits for detect(), candidate lists, counters
    size_t off = roundup_sz(net->act_bytes, 4096);
    net->logits_off = off;
    off += roundup_sz(net->out_count * 4 * (size_t)net->opt.max_batch, 4096);
    net->cand_off = off;
    off += roundup_sz(sizeof(Candidate) * (size_t)net->opt.cand_capacity * net->opt.max_batch, 4096);
    net->count_off = off;
    off += roundup_sz(sizeof(int) * (size_t)net->opt.max_batch * kCandCountStride, 4096);
    net->nms_off = off;
    off += roundup_sz(nms_scratch_bytes(net->opt.cand_capacity) * (size_t)net->opt.max_batch, 4096);
    net->obj_off = off;
    net->obj_bytes = net->head.n_classes > 0 ? (size_t)net->opt.max_batch * (net->out_count / (size_t)(5 + net->head.n_classes)) * 4 : 0;
    off += roundup_sz(net->obj_bytes, 4096);
    // split-K slabs (float32 partial sums of the convs whose launch would leave the chip idle): last region of the workspace,
    // sized by yolo_net_create from the launches that can actually split (api.cpp: splitk_slab_bytes) -- 0 for most big-batch nets
    net->splitk_off = off;
    net->splitk_bytes = 0;
    net->workspace_bytes = off;
    return YOLO_OK;
}

// Darknet stream -> device layout.  Order per conv (net/layers.py:53-63; net/base.py:26-46):
// BN: beta, gamma, moving_mean, moving_variance, kernel[out][in][kh][kw]; else bias, kernel.
// Fold: w' = w * gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps), eps = 1e-5
// (net/layers.py:5), computed in float64 and rounded once.
int pack_weights(const yolo_net *net, const float *host, size_t n, std::vector<unsigned char> &blob, std::string &err) {
    if (n != net->weight_count) {
        err = "weight stream holds " + std::to_string(n) + " values, the layer list needs " + std::to_string(net->weight_count);
        return YOLO_ERR_WEIGHTS;
    }
    blob.assign(net->weights_bytes, 0);
    const bool f16 = net->opt.dtype == YOLO_DTYPE_F16;
    const int epc = net->epc;
    for (const Kernel &k : net->kernels) {
        if (k.kind == K_FIRST) {        // [27 = (kh,kw,cin)][cout] float32 (first.hip)
            const float *p = host + k.w_src;
            const float *beta = nullptr, *gamma = nullptr, *mean = nullptr, *var = nullptr, *bias = nullptr;
            if (k.batch_norm) { beta = p; gamma = p + k.cout; mean = p + 2 * k.cout; var = p + 3 * k.cout; p += 4 * (size_t)k.cout; }
            else { bias = p; p += k.cout; }
            float *wdst = reinterpret_cast<float *>(blob.data() + k.w_off);
            float *bdst = reinterpret_cast<float *>(blob.data() + k.b_off);
            for (int o = 0; o < k.cout; ++o) {
                double scale = 1.0;
                if (k.batch_norm) {
                    scale = (double)gamma[o] / std::sqrt((double)var[o] + 1e-5);
                    bdst[o] = (float)((double)beta[o] - (double)mean[o] * scale);
                } else {
                    bdst[o] = bias[o];
                }
                for (int t = 0; t < 9; ++t)
                    for (int ci = 0; ci < 3; ++ci) {
                        float v = (float)((double)p[((size_t)o * 3 + ci) * 9 + t] * scale);
                        if (f16) v = (float)(_Float16)v;        // same operand rounding as the MFMA path
                        wdst[(t * 3 + ci) * k.cout + o] = v;
                    }
            }
            continue;
        }
        if (k.kind != K_CONV) continue;
        const int taps = k.ksize * k.ksize;
        const float *p = host + k.w_src;
        const float *beta = nullptr, *gamma = nullptr, *mean = nullptr, *var = nullptr, *bias = nullptr;
        if (k.batch_norm) { beta = p; gamma = p + k.cout; mean = p + 2 * k.cout; var = p + 3 * k.cout; p += 4 * (size_t)k.cout; }
        else { bias = p; p += k.cout; }
        const float *kern = p;      // [out][in][kh][kw]
        float *bdst = reinterpret_cast<float *>(blob.data() + k.b_off);
        const size_t wrow = (size_t)k.ktiles * 128;
        for (int o = 0; o < k.cout; ++o) {
            double scale = 1.0;
            if (k.batch_norm) {
                scale = (double)gamma[o] / std::sqrt((double)var[o] + 1e-5);
                bdst[o] = (float)((double)beta[o] - (double)mean[o] * scale);
            } else {
                bdst[o] = bias[o];
            }
            unsigned char *row = blob.data() + k.w_off + (size_t)o * wrow;
            for (int t = 0; t < taps; ++t) {
                for (int ci = 0; ci < k.cin; ++ci) {
                    float v = (float)((double)kern[((size_t)o * k.cin + ci) * taps + t] * scale);
                    size_t e = (size_t)t * k.cin_s + ci;     // (kh,kw) major, cin minor; chunk = e / epc
                    if (f16) reinterpret_cast<_Float16 *>(row)[e] = (_Float16)v;
                    else reinterpret_cast<float *>(row)[e] = v;
                }
            }
        }
        (void)epc;
    }
    return YOLO_OK;
}

static const char *kind_name(int k) {
    switch (k) { case K_PREP: return "prep"; case K_CONV: return "conv"; case K_POOL: return "maxpool"; case K_FIRST: return "conv_first"; default: return "eltwise"; }
}

std::string describe(const yolo_net *net) {
    std::ostringstream o;
    o << "yolo_hip plan: dtype=" << (net->opt.dtype == YOLO_DTYPE_F16 ? "f16" : "f32") << " max_batch=" << net->opt.max_batch
      << " layers=" << net->layers.size() << " kernels=" << net->kernels.size() << " buffers=" << net->buffers.size() << "\n";
    o << "  weights: " << net->weight_count << " floats -> " << net->weights_bytes << " B packed; activations "
      << net->act_bytes << " B; workspace " << net->workspace_bytes << " B; GFLOP/image " << net->flops_per_image * 1e-9 << "\n";
    int idx = 0;
    for (const Kernel &k : net->kernels) {
        o << "  [" << idx++ << "] " << kind_name(k.kind) << " layer " << k.layer;
        if (k.kind == K_CONV)
            o << " (conv@" << k.src_layer << ") " << k.ksize << "x" << k.ksize << "/" << k.stride << " " << k.cin << "->" << k.cout
              << " cfg=N" << (k.cfg == CFG_N128 ? 128 : k.cfg == CFG_N64 ? 64 : 32) << (k.perchunk ? " perchunk" : "")
              << " ktiles=" << k.ktiles;
        o << " in=b" << k.in.buf << "[" << k.in.H << "x" << k.in.W << "x" << k.in.C << " ld" << k.in.ld << "+" << k.in.coff << "]";
        o << " out=b" << k.out.buf << "[" << k.out.H << "x" << k.out.W << "x" << k.out.C << " ld" << k.out.ld << "+" << k.out.coff
          << (k.out.f32 ? " f32" : "") << "]";
        if (!k.note.empty()) o << " " << k.note;
        o << "\n";
    }
    return o.str();
}

}  // namespace yolo
