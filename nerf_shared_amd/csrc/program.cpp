// program.cpp -- host-side construction of the fragment programs (see program.h).
#include "program.h"

#include <algorithm>

namespace na {

namespace {
struct Seg { int kind, col_base, len, nk, L; };
}

int build_program(const nerf_amd_arch &a, Program &p, const char **err) {
    p = Program();
    p.arch = a;
    // any width the reference constructor takes, up to 1024: the fp32 kernel's 32-row tiles and 8-column groups are
    // guarded / zero-padded (mlp_fp32.hip), W / 2 is the reference's W // 2 (nerf.py:83)
    if (a.D < 1 || a.D > 64 || a.W < 2 || a.W > 1024) { *err = "D/W out of range (D 1..64, W 2..1024)"; return -1; }
    if (a.n_skips < 0 || a.n_skips > NERF_AMD_MAX_SKIPS) { *err = "too many skips"; return -1; }
    if (a.i_embed != 0 && a.i_embed != -1) { *err = "i_embed must be 0 or -1"; return -1; }
    if (a.multires < 0 || a.multires > 20 || a.multires_views < 0 || a.multires_views > 20) { *err = "multires out of range"; return -1; }
    const int Lx = a.i_embed == -1 ? 0 : a.multires;
    const int Ld = a.i_embed == -1 ? 0 : a.multires_views;
    p.input_ch = embed_dim(a.multires, a.i_embed);
    p.input_ch_views = a.use_viewdirs ? embed_dim(a.multires_views, a.i_embed) : 0;
    p.out_ch = a.use_viewdirs ? 4 : a.output_ch;
    if (p.out_ch < 1 || p.out_ch > 32) { *err = "output_ch must be in 1..32"; return -1; }
    auto is_skip = [&](int i) { for (int s = 0; s < a.n_skips; ++s) if (a.skips[s] == i) return true; return false; };
    if (is_skip(a.D - 1)) { *err = "a skip on the last pts_linear feeds W+input_ch columns into a W-column head (the reference fails too)"; return -1; }
    const int W = a.W, D = a.D;

    // ---- tensors in nerf_amd.h order
    for (int i = 0; i < D; ++i) {
        int n_in = i == 0 ? p.input_ch : (is_skip(i - 1) ? W + p.input_ch : W);
        p.tensors.push_back({W, n_in});
    }
    if (a.use_viewdirs) {
        p.tensors.push_back({W, W});                          // feature_linear
        p.tensors.push_back({1, W});                          // alpha_linear
        p.tensors.push_back({W / 2, W + p.input_ch_views});   // views_linears.0
        p.tensors.push_back({3, W / 2});                      // rgb_linear
    } else {
        p.tensors.push_back({a.output_ch, W});                // output_linear
    }

    // ---- fp32 generic program
    {
        p.lds_rows = (p.input_ch + W + p.input_ch_views + 8 + 1) & ~1;   // +8: k-groups of 8 may read past the last valid row
        // the fp32 kernel keeps every feature row of 64 (or, for the widest models, 32) points in the CU's 160 KiB of LDS
        if ((int64_t)p.lds_rows * 32 * 4 > 160 * 1024) { *err = "W + encoding widths exceed the 1280 feature rows the exact kernel can hold in LDS"; return -1; }
        int cur = 0;
        int64_t foff = 0, boff = 0;
        auto push = [&](int tensor, int in_row, int out_row, int out_col, int relu, bool flip) {
            LayerF32 l;
            l.tensor = tensor; l.n_out = p.tensors[tensor].n_out; l.n_in = p.tensors[tensor].n_in;
            l.in_row = in_row; l.out_row = out_row; l.out_col = out_col; l.relu = relu; l.in_buf = cur;
            l.frag_off = foff; l.bias_off = boff;
            int tiles = (l.n_out + 31) / 32, groups = (l.n_in + 7) / 8;
            foff += (int64_t)tiles * groups * 256;
            boff += tiles * 32;
            p.layers.push_back(l);
            if (flip) cur ^= 1;
        };
        for (int i = 0; i < D; ++i) {
            int in_row = (i == 0 || is_skip(i - 1)) ? 0 : p.input_ch;
            push(i, in_row, p.input_ch, 0, 1, true);
        }
        if (a.use_viewdirs) {
            push(D + 1, p.input_ch, -1, 3, 0, false);          // alpha -> out[:,3]
            push(D + 0, p.input_ch, p.input_ch, 0, 0, true);   // feature
            push(D + 2, p.input_ch, p.input_ch, 0, 1, true);   // views
            push(D + 3, p.input_ch, -1, 0, 0, false);          // rgb -> out[:,0:3]
        } else {
            push(D, p.input_ch, -1, 0, 0, false);
        }
        p.f32_stream_floats = foff;
        p.f32_bias_floats = boff;
        // ---- exact-fp32 training: workspace rows, transposed fragments, who accumulates
        {
            const int ic = p.input_ch;
            int64_t toff = 0;
            int rows = 0;
            p.lds_rows_bwd = p.lds_rows + 32;
            // the value a layer reads from the hidden rows is identified by the last layer that wrote them (-1: none)
            int writer = -1;
            std::vector<int> in_value(p.layers.size());
            for (size_t l = 0; l < p.layers.size(); ++l) {
                in_value[l] = writer;
                if (p.layers[l].out_row >= 0) writer = (int)l;
            }
            for (size_t l = 0; l < p.layers.size(); ++l) {
                const LayerF32 &L = p.layers[l];
                TrainLayerF32 t;
                t.frag_off_t = toff;
                toff += (int64_t)((L.n_in + 31) / 32) * ((L.n_out + 7) / 8) * 256;
                t.x_row = rows; rows += L.n_in;
                t.g_row = rows; rows += L.n_out;
                t.y_row = L.relu ? rows : -1;
                if (L.relu) rows += L.n_out;
                // hidden rows of the input: LDS rows [ic, ic + W) (what a previous layer produced), relative to in_row
                const int a0 = std::max(L.in_row, ic), a1 = std::min(L.in_row + L.n_in, ic + W);
                t.lo = a1 > a0 && in_value[l] >= 0 ? a0 - L.in_row : 0;
                t.hi = a1 > a0 && in_value[l] >= 0 ? a1 - L.in_row : 0;
                t.accumulate = 0;
                for (size_t j = l + 1; j < p.layers.size(); ++j)     // processed before l in the backward order
                    if (in_value[j] == in_value[l] && in_value[l] >= 0) t.accumulate = 1;
                t.lds_g_row = L.out_row >= 0 ? L.out_row : p.lds_rows + L.out_col;
                p.tlayers.push_back(t);
            }
            p.f32_stream_t_floats = toff;
            p.train_f32_rows = rows;
        }
    }

    // ---- bf16 fused program: canonical D=8, W=256, skips=[4]
    p.bf16_ok = (D == 8 && W == 256 && a.n_skips == 1 && a.skips[0] == 4);
    if (p.bf16_ok) {
        p.KE = gen_ksteps(Lx);
        p.KD = a.use_viewdirs ? gen_ksteps(Ld) : 0;
        auto add_layer = [&](int tensor, std::initializer_list<Seg> segs) {
            int n_out = p.tensors[tensor].n_out;
            for (int t = 0; t < (n_out + 31) / 32; ++t) {
                p.tiles.push_back({tensor, 32 * t});
                for (const Seg &s : segs)
                    for (int ks = 0; ks < s.nk; ++ks)
                        p.frags.push_back({tensor, s.kind, 32 * t, s.col_base, ks, s.len, s.L, 0});
            }
        };
        const Seg E{FRAG_GEN, 0, p.input_ch, p.KE, Lx};
        const Seg H{FRAG_ACC, 0, W, 16, 0};
        add_layer(0, {E});
        for (int i = 1; i <= 4; ++i) add_layer(i, {H});
        add_layer(5, {E, Seg{FRAG_ACC, p.input_ch, W, 16, 0}});
        add_layer(6, {H});
        add_layer(7, {H});
        if (a.use_viewdirs) {
            add_layer(8, {H});                                                         // feature
            add_layer(9, {H});                                                         // alpha (1 row)
            add_layer(10, {H, Seg{FRAG_GEN, W, p.input_ch_views, p.KD, Ld}});          // views
            add_layer(11, {Seg{FRAG_ACC, 0, W / 2, 8, 0}});                            // rgb (3 rows)
        } else {
            add_layer(8, {H});                                                         // output_linear
        }
        p.n_frags_used = (int)p.frags.size();
        while (p.frags.size() % STREAM_PAD_FRAGS) p.frags.push_back({0, FRAG_ZERO, 0, 0, 0, 0, 0, 0});

        // ---- s16 program: layer -> pair of 16-row tiles -> k-step -> tile of the pair
        p.KE16 = gen16_ksteps(Lx);
        p.KD16 = a.use_viewdirs ? gen16_ksteps(Ld) : 0;
        auto add_layer16 = [&](int tensor, std::initializer_list<Seg> segs) {
            const int n_out = p.tensors[tensor].n_out;
            const int n_tiles = (n_out + 15) / 16;
            for (int t = 0; t < n_tiles; t += 2) {
                const int in_pair = (t + 1 < n_tiles) ? 2 : 1;
                for (int u = 0; u < in_pair; ++u) p.tiles16.push_back({tensor, 16 * (t + u)});
                for (const Seg &s : segs)
                    for (int ks = 0; ks < s.nk; ++ks) {
                        for (int u = 0; u < in_pair; ++u)
                            p.frags16.push_back({tensor, s.kind, 16 * (t + u), s.col_base, ks, s.len, s.L, 0});
                        for (int part = 1; part <= 2; ++part)          // the same fragments as fp16 hi / lo parts
                            for (int u = 0; u < in_pair; ++u)
                                p.frags_split.push_back({tensor, s.kind, 16 * (t + u), s.col_base, ks, s.len, s.L, part});
                    }
            }
        };
        const Seg E16{FRAG_GEN16, 0, p.input_ch, p.KE16, Lx};
        const Seg H16{FRAG_ACC16, 0, W, 8, 0};
        add_layer16(0, {E16});
        for (int i = 1; i <= 4; ++i) add_layer16(i, {H16});
        add_layer16(5, {E16, Seg{FRAG_ACC16, p.input_ch, W, 8, 0}});
        add_layer16(6, {H16});
        add_layer16(7, {H16});
        if (a.use_viewdirs) {
            add_layer16(8, {H16});
            add_layer16(9, {H16});
            add_layer16(10, {H16, Seg{FRAG_GEN16, W, p.input_ch_views, p.KD16, Ld}});
            add_layer16(11, {Seg{FRAG_ACC16, 0, W / 2, 4, 0}});
        } else {
            add_layer16(8, {H16});
        }
        p.n_frags16_used = (int)p.frags16.size();
        while (p.frags16.size() % STREAM_PAD_FRAGS) p.frags16.push_back({0, FRAG_ZERO, 0, 0, 0, 0, 0, 0});
        p.n_frags_split_used = (int)p.frags_split.size();
        while (p.frags_split.size() % STREAM_PAD_FRAGS) p.frags_split.push_back({0, FRAG_ZERO, 0, 0, 0, 0, 0, 0});

        // ---- backward stream: g_in^T = W^T g_out^T, layer -> pair of 16-row input-feature tiles ->
        //      segment -> k-step -> tile of the pair (parameter gradients only: the encodings' own
        //      gradients, i.e. ray gradients, are not propagated)
        {
            struct SegT { int tensor, kind, col_base, nk; };
            auto add_bwd = [&](int n_in, std::initializer_list<SegT> segs) {
                for (int t = 0; t < n_in / 16; t += 2)
                    for (const SegT &sg : segs)
                        for (int ks = 0; ks < sg.nk; ++ks) {
                            for (int u = 0; u < 2; ++u)
                                p.frags_bwd.push_back({sg.tensor, sg.kind, 16 * (t + u), sg.col_base, ks,
                                                       p.tensors[sg.tensor].n_out, n_in, 0});
                            for (int part = 1; part <= 2; ++part)
                                for (int u = 0; u < 2; ++u)
                                    p.frags_bwd_split.push_back({sg.tensor, sg.kind, 16 * (t + u), sg.col_base, ks,
                                                                 p.tensors[sg.tensor].n_out, n_in, part});
                        }
            };
            // encoding-slot rows (ray gradients): n_in = 32 * k-steps of the encoding, L = its multires
            auto add_bwd_enc = [&](int tensor, int col_base, int n_slots, int L, int nk) {
                for (int t = 0; t < n_slots / 16; t += 2)
                    for (int ks = 0; ks < nk; ++ks) {
                        for (int u = 0; u < 2; ++u)
                            p.frags_bwd.push_back({tensor, FRAG_TE16, 16 * (t + u), col_base, ks, p.tensors[tensor].n_out, L, 0});
                        for (int part = 1; part <= 2; ++part)
                            for (int u = 0; u < 2; ++u)
                                p.frags_bwd_split.push_back({tensor, FRAG_TE16, 16 * (t + u), col_base, ks, p.tensors[tensor].n_out, L, part});
                    }
            };
            if (a.use_viewdirs) {
                add_bwd(W / 2, {SegT{D + 3, FRAG_TG16, 0, 1}});                                // g_hv   <- rgb_linear
                add_bwd(W, {SegT{D + 2, FRAG_T16, 0, (W / 2) / 32}});                          // g_feat <- views_linears.0
                add_bwd_enc(D + 2, W, 32 * p.KD16, Ld, (W / 2) / 32);                          // g_dirs <- views_linears.0[:, W:]
                add_bwd(W, {SegT{D + 0, FRAG_T16, 0, W / 32}, SegT{D + 1, FRAG_TG16, 0, 1}});  // g_h8   <- feature + alpha
            } else {
                add_bwd(W, {SegT{D, FRAG_TG16, 0, 1}});                                        // g_h8   <- output_linear (<= 16 rows)
            }
            for (int l = D - 1; l >= 1; --l) {                                                 // g_h(l) <- pts_linears.l
                add_bwd(W, {SegT{l, FRAG_T16, is_skip(l - 1) ? p.input_ch : 0, W / 32}});
                if (is_skip(l - 1)) add_bwd_enc(l, 0, 32 * p.KE16, Lx, W / 32);                // g_e    <- its [input_pts] columns
            }
            add_bwd_enc(0, 0, 32 * p.KE16, Lx, W / 32);                                        // g_e    <- pts_linears.0
            p.n_frags_bwd_used = (int)p.frags_bwd.size();
            while (p.frags_bwd.size() % STREAM_PAD_FRAGS) p.frags_bwd.push_back({0, FRAG_ZERO, 0, 0, 0, 0, 0, 0});
            p.n_frags_bwd_split_used = (int)p.frags_bwd_split.size();
            while (p.frags_bwd_split.size() % STREAM_PAD_FRAGS) p.frags_bwd_split.push_back({0, FRAG_ZERO, 0, 0, 0, 0, 0, 0});
        }
    }
    return 0;
}

}  // namespace na
