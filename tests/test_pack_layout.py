"""CPU check of the bf16 fragment stream (program.h / pack.hip) without a GPU.

A numpy emulator replays mlp_bf16.hip's data flow lane by lane -- MFMA operand
maps of v_mfma_f32_32x32x16_bf16, accumulator-as-next-operand, generated
encodings on lane halves, bias table, layer order -- on the stream produced by
the library's host packer (nerf_amd_pack_bf16_host, the twin of the device
pack kernel).  If the emulated network agrees with a plain bf16-rounded
evaluation of the same weights, the permutations baked into the stream are
consistent with the kernel's register layout.
"""
import ctypes

import numpy as np
import pytest
import torch

from nerf_shared_amd import _lib, synth
from oracle import nerf_oracle as O

lib = _lib.lib


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (numpy)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def gen_col(ks, h, j, L):
    e = 8 * ks + j
    if e < 3 * L:
        return 3 + 6 * (e // 3) + 3 * h + (e % 3)
    if e == 3 * L:
        return 2 if h else 0
    if e == 3 * L + 1:
        return -1 if h else 1
    return -1


def gen_ksteps(L):
    return (3 * L + 2 + 7) // 8


def host_pack(arch_kwargs, sd, shape=32):
    names = ["pts_linears.%d" % i for i in range(arch_kwargs["D"])]
    if arch_kwargs["use_viewdirs"]:
        names += ["feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"]
    else:
        names += ["output_linear"]
    ws = [np.ascontiguousarray(sd[n + ".weight"], np.float32) for n in names]
    bs = [np.ascontiguousarray(sd[n + ".bias"], np.float32) for n in names]
    arch = _lib.make_arch(arch_kwargs["D"], arch_kwargs["W"], arch_kwargs["output_ch"], arch_kwargs["skips"],
                          arch_kwargs["use_viewdirs"], arch_kwargs["multires"], arch_kwargs["multires_views"], 0)
    n = len(names)
    wp = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    nf, nb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.nerf_amd_pack_bf16_host(ctypes.byref(arch), shape, wp, bp, n, None, ctypes.byref(nf), None, ctypes.byref(nb)), "size query")
    stream = np.zeros(nf.value * 512, np.uint16)
    bias = np.zeros(nb.value, np.float32)
    _lib.check(lib.nerf_amd_pack_bf16_host(ctypes.byref(arch), shape, wp, bp, n,
                                           stream.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), ctypes.byref(nf),
                                           bias.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(nb)), "pack")
    if shape == 16:
        return bf16_bits_to_f32(stream).reshape(nf.value, 64, 8), bias.reshape(-1, 16)
    return bf16_bits_to_f32(stream).reshape(nf.value, 64, 8), bias.reshape(-1, 2, 16)


# ---------------------------------------------------------------- 16x16x32 ("s16") twin of program.h
def gen16_ksteps(L):
    return (3 * L + 2 + 15) // 16


def gen16_ntrig(L, b):
    return 3 * ((L - b + 1) // 2)


def gen16_misc(L, h, b, m):
    cap = 8 * gen16_ksteps(L)
    g = h if b else 2 + h
    offset = sum(cap - gen16_ntrig(L, 1 if gg < 2 else 0) for gg in range(g))
    idx = offset + m
    return idx if idx < 3 else -1


def gen16_col(ks, q, j, L):
    h, b, i = q >> 1, q & 1, 8 * ks + j
    n = gen16_ntrig(L, b)
    if i < n:
        return 3 + 6 * (2 * (i // 3) + b) + 3 * h + (i % 3)
    return gen16_misc(L, h, b, i - n)


class WaveEmu16:
    """One wave = 32 points as two 16-column tiles, walked as mlp_bf16_s16.hip does."""

    def __init__(self, stream, bias):
        self.stream, self.bias = stream, bias

    @staticmethod
    def mfma(a_frag, b_frag, acc):
        # A[row = l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col = l&15]; D: col = l&15, row = 4*(l>>4)+r
        A = np.zeros((16, 32), np.float32)
        B = np.zeros((32, 16), np.float32)
        for l in range(64):
            A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a_frag[l]
            B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b_frag[l]
        D = A.astype(np.float64) @ B.astype(np.float64)
        out = acc.copy()
        for l in range(64):
            for r in range(4):
                out[l, r] += D[4 * (l >> 4) + r, l & 15]
        return out

    def bias_init(self, t):
        acc = np.zeros((64, 4), np.float64)
        for l in range(64):
            acc[l] = self.bias[t, 4 * (l >> 4):4 * (l >> 4) + 4]
        return acc

    def tile_pair(self, f0, t, x1, k1, x2=None, k2=0):
        acc = [[self.bias_init(t), self.bias_init(t)], [self.bias_init(t + 1), self.bias_init(t + 1)]]
        n = f0
        for xs, kk in ((x1, k1), (x2, k2)):
            for k in range(kk):
                for u in range(2):
                    w = self.stream[n]
                    n += 1
                    for cc in range(2):
                        acc[u][cc] = self.mfma(w, xs[2 * k + cc], acc[u][cc])
        return acc

    def tile_single(self, f0, t, x1, k1):
        acc = [self.bias_init(t), self.bias_init(t)]
        for k in range(k1):
            for cc in range(2):
                acc[cc] = self.mfma(self.stream[f0 + k], x1[2 * k + cc], acc[cc])
        return [a.astype(np.float32) for a in acc]

    def layer(self, f0, t0, npair, x1, k1, x2=None, k2=0, relu=True):
        y = []
        for p in range(npair):
            acc = self.tile_pair(f0 + p * 2 * (k1 + k2), t0 + 2 * p, x1, k1, x2, k2)
            for cc in range(2):
                ev, od = acc[0][cc].astype(np.float32), acc[1][cc].astype(np.float32)
                if relu:
                    ev, od = np.maximum(ev, 0), np.maximum(od, 0)
                y.append(bf16_round(np.concatenate([ev, od], -1)))
        return y

    @staticmethod
    def encode(emb, L, K):
        """emb [32, 3+6L] -> fragments [2*K] of [64, 8]: index 2*ks + cc (FRAG_GEN16 order)."""
        out = []
        for ks in range(K):
            for cc in range(2):
                f = np.zeros((64, 8), np.float32)
                for l in range(64):
                    for j in range(8):
                        c = gen16_col(ks, l >> 4, j, L)
                        f[l, j] = emb[16 * cc + (l & 15), c] if c >= 0 else 0.0
                out.append(bf16_round(f))
        return out


def emulate16(arch, sd, pts, dirs, keep=None):
    """`keep` (dict) receives the fragment lists of every layer output (the training forward's saves)."""
    LX, LD, VD = arch["multires"], arch["multires_views"], arch["use_viewdirs"]
    KE, KD = gen16_ksteps(LX), (gen16_ksteps(LD) if VD else 0)
    stream, bias = host_pack(arch, sd, 16)
    F_L1 = 16 * KE
    F_L5 = F_L1 + 4 * 128
    F_L6 = F_L5 + 16 * (KE + 8)
    F_HEAD = F_L6 + 256
    w = WaveEmu16(stream, bias)
    E = w.encode(O.embed(torch.from_numpy(pts), LX).numpy(), LX, KE)
    hs = []
    A = w.layer(0, 0, 8, E, KE); hs.append(A)
    B = w.layer(F_L1, 16, 8, A, 8); hs.append(B)
    A = w.layer(F_L1 + 128, 32, 8, B, 8); hs.append(A)
    B = w.layer(F_L1 + 256, 48, 8, A, 8); hs.append(B)
    A = w.layer(F_L1 + 384, 64, 8, B, 8); hs.append(A)
    B = w.layer(F_L5, 80, 8, E, KE, A, 8); hs.append(B)
    A = w.layer(F_L6, 96, 8, B, 8); hs.append(A)
    B = w.layer(F_L6 + 128, 112, 8, A, 8); hs.append(B)
    if keep is not None:
        keep["h"] = hs
    out = np.zeros((32, 4 if VD else arch["output_ch"]), np.float32)
    if VD:
        Dv = w.encode(O.embed(torch.from_numpy(dirs), LD).numpy(), LD, KD)
        F_ALPHA = F_HEAD + 128
        F_VIEWS = F_ALPHA + 8
        F_RGB = F_VIEWS + 8 * (8 + KD)
        assert F_RGB + 4 <= stream.shape[0]
        A = w.layer(F_HEAD, 128, 8, B, 8, relu=False)
        alpha = w.tile_single(F_ALPHA, 144, B, 8)
        B2 = w.layer(F_VIEWS, 145, 4, A, 8, Dv, KD)
        if keep is not None:
            keep["feat"], keep["hv"] = A, B2
        rgb = w.tile_single(F_RGB, 153, B2, 4)
        for cc in range(2):
            out[16 * cc:16 * cc + 16, 0:3] = rgb[cc][:16, 0:3]
            out[16 * cc:16 * cc + 16, 3] = alpha[cc][:16, 0]
        return out
    o = w.tile_single(F_HEAD, 128, B, 8)
    for cc in range(2):
        for qq in range(4):
            for r in range(4):
                row = 4 * qq + r
                if row < arch["output_ch"]:
                    out[16 * cc:16 * cc + 16, row] = o[cc][16 * qq:16 * qq + 16, r]
    return out


class WaveEmu:
    """One wave = 32 points, emulated exactly as mlp_bf16.hip walks the stream."""

    def __init__(self, stream, bias):
        self.stream, self.bias = stream, bias
        self.lane = np.arange(64)
        self.col, self.h = self.lane & 31, self.lane >> 5

    def mfma(self, a_frag, b_frag, acc):
        # A[row = l&31][k = 8*(l>>5)+j], B[k = 8*(l>>5)+j][col = l&31]
        A = np.zeros((32, 16), np.float32)
        B = np.zeros((16, 32), np.float32)
        for l in range(64):
            A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a_frag[l]
            B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b_frag[l]
        D = A.astype(np.float64) @ B.astype(np.float64)
        out = acc.copy()
        for l in range(64):
            for r in range(16):
                out[l, r] += D[acc_row(r, l >> 5), l & 31]
        return out

    def tile(self, f0, t, x1, k1, x2=None, k2=0):
        acc = np.zeros((64, 16), np.float64)
        for l in range(64):
            acc[l] = self.bias[t, l >> 5]
        for k in range(k1):
            acc = self.mfma(self.stream[f0 + k], x1[k], acc)
        for k in range(k2):
            acc = self.mfma(self.stream[f0 + k1 + k], x2[k], acc)
        return acc.astype(np.float32)

    def layer(self, f0, t0, nt, x1, k1, x2=None, k2=0, relu=True):
        y = []
        for t in range(nt):
            acc = self.tile(f0 + t * (k1 + k2), t0 + t, x1, k1, x2, k2)
            if relu:
                acc = np.maximum(acc, 0)
            y.append(bf16_round(acc[:, 0:8]))
            y.append(bf16_round(acc[:, 8:16]))
        return y

    def encode(self, emb, L, K):
        """emb [32, 3+6L] fp32 reference embedding -> K fragments [64, 8] (FRAG_GEN order)."""
        out = []
        for ks in range(K):
            f = np.zeros((64, 8), np.float32)
            for l in range(64):
                for j in range(8):
                    c = gen_col(ks, l >> 5, j, L)
                    f[l, j] = emb[l & 31, c] if c >= 0 else 0.0
            out.append(bf16_round(f))
        return out


def emulate(arch, sd, pts, dirs):
    LX, LD, VD = arch["multires"], arch["multires_views"], arch["use_viewdirs"]
    KE, KD = gen_ksteps(LX), (gen_ksteps(LD) if VD else 0)
    stream, bias = host_pack(arch, sd)
    F_L1 = 8 * KE
    F_L5 = F_L1 + 4 * 128
    F_L6 = F_L5 + 8 * (KE + 16)
    F_HEAD = F_L6 + 256
    w = WaveEmu(stream, bias)
    E = w.encode(O.embed(torch.from_numpy(pts), LX).numpy(), LX, KE)
    A = w.layer(0, 0, 8, E, KE)
    B = w.layer(F_L1, 8, 8, A, 16)
    A = w.layer(F_L1 + 128, 16, 8, B, 16)
    B = w.layer(F_L1 + 256, 24, 8, A, 16)
    A = w.layer(F_L1 + 384, 32, 8, B, 16)
    B = w.layer(F_L5, 40, 8, E, KE, A, 16)
    A = w.layer(F_L6, 48, 8, B, 16)
    B = w.layer(F_L6 + 128, 56, 8, A, 16)
    if VD:
        Dv = w.encode(O.embed(torch.from_numpy(dirs), LD).numpy(), LD, KD)
        F_FEAT, F_ALPHA = F_HEAD, F_HEAD + 128
        F_VIEWS = F_ALPHA + 16
        F_RGB = F_VIEWS + 4 * (16 + KD)
        assert F_RGB + 8 <= stream.shape[0]
        A = w.layer(F_FEAT, 64, 8, B, 16, relu=False)
        alpha = w.tile(F_ALPHA, 72, B, 16)
        B2 = w.layer(F_VIEWS, 73, 4, A, 16, Dv, KD)
        rgb = w.tile(F_RGB, 77, B2, 8)
        return np.stack([rgb[:32, 0], rgb[:32, 1], rgb[:32, 2], alpha[:32, 0]], -1)
    o = w.tile(F_HEAD, 64, B, 16)
    out = np.zeros((32, arch["output_ch"]), np.float32)
    for r in range(16):
        for h in range(2):
            row = acc_row(r, h)
            if row < arch["output_ch"]:
                out[:, row] = o[32 * h:32 * h + 32, r]
    return out


def plain_bf16(arch, sd, pts, dirs):
    """Same network, natural layout: weights and every activation rounded to bf16, fp64 accumulate."""
    W = {k: bf16_round(v).astype(np.float64) if k.endswith("weight") else v.astype(np.float64) for k, v in sd.items()}
    x = bf16_round(O.embed(torch.from_numpy(pts), arch["multires"]).numpy()).astype(np.float64)
    lin = lambda n, v: v @ W[n + ".weight"].T + W[n + ".bias"]       # noqa: E731
    rb = lambda v: bf16_round(v.astype(np.float32)).astype(np.float64)  # noqa: E731
    h = x
    for i in range(8):
        h = rb(np.maximum(lin("pts_linears.%d" % i, h), 0))
        if i == 4:
            h = np.concatenate([x, h], -1)
    if not arch["use_viewdirs"]:
        return lin("output_linear", h)
    d = bf16_round(O.embed(torch.from_numpy(dirs), arch["multires_views"]).numpy()).astype(np.float64)
    sigma = lin("alpha_linear", h)
    feat = rb(lin("feature_linear", h))
    hv = rb(np.maximum(lin("views_linears.0", np.concatenate([feat, d], -1)), 0))
    return np.concatenate([lin("rgb_linear", hv), sigma], -1)


CASES = [
    dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=10, multires_views=4),
    dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=True, multires=15, multires_views=6),
    dict(D=8, W=256, output_ch=5, skips=[4], use_viewdirs=False, multires=10, multires_views=4),
]


@pytest.mark.parametrize("arch", CASES, ids=["vd_10_4", "vd_15_6", "novd_10"])
def test_s16_stream_matches_kernel_dataflow(arch):
    """The 16x16x32 stream against the same plain evaluation (and so against the 32x32x16 one)."""
    rng = np.random.default_rng(7)
    pts = rng.uniform(-3, 3, size=(32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    sd = synth.make_state_dict(3, 3.0, **{**arch, "skips": tuple(arch["skips"])})
    np.testing.assert_allclose(emulate16(arch, sd, pts, dirs), plain_bf16(arch, sd, pts, dirs), atol=2e-3, rtol=2e-3)


def frags_to_rows(frags, n_ks):
    """[2*n_ks] fragments [64, 8] -> natural-order rows [32 points, 32*n_ks features] (acc16 slot order undone)."""
    rows = np.zeros((32, 32 * n_ks), np.float32)
    for ks in range(n_ks):
        for cc in range(2):
            f = frags[2 * ks + cc]
            for l in range(64):
                q = l >> 4
                for j in range(8):
                    rows[16 * cc + (l & 15), 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3)] = f[l, j]
    return rows


@pytest.mark.parametrize("arch", CASES[:2], ids=["vd_10_4", "vd_15_6"])
def test_backward_stream_matches_kernel_dataflow(arch):
    """mlp_bwd_s16.hip's dX chain replayed on the host-packed transposed stream (shape 17): the
    pre-activation gradients of every layer against a plain numpy backward with the same roundings,
    for both encodings the training kernels are instantiated for."""
    LX, LD = arch["multires"], arch["multires_views"]
    KE, KD = gen16_ksteps(LX), gen16_ksteps(LD)
    ic, icv = 3 + 6 * LX, 3 + 6 * LD
    rng = np.random.default_rng(9)
    pts = rng.uniform(-3, 3, size=(32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    sd = synth.make_state_dict(3, 2.0, **{**arch, "skips": tuple(arch["skips"])})
    keep = {}
    emulate16(arch, sd, pts, dirs, keep)
    g_raw = rng.normal(size=(32, 4)).astype(np.float32)
    stream, _ = host_pack(arch, sd, 17)
    assert stream.shape[0] % 192 == 0
    w = WaveEmu16(stream, np.zeros((400, 16), np.float32))

    def gen_frag(cols):                      # FRAG_TG16 operand: k slot (q=0, j) = column j of g_raw
        out = []
        for cc in range(2):
            f = np.zeros((64, 8), np.float32)
            for l in range(16):
                for j, c in enumerate(cols):
                    f[l, j] = g_raw[16 * cc + l, c]
            out.append(bf16_round(f))
        return out

    def tlayer(f0, npair, x1, k1, x2=None, k2=0, mask=None):
        g = w.layer(f0, 0, npair, x1, k1, x2, k2, relu=False)       # bias table is zero here
        if mask is not None:
            g = [np.where(m != 0, v, 0).astype(np.float32) for v, m in zip(g, mask)]
        return g

    Grgb, Gsig = gen_frag([0, 1, 2]), gen_frag([3])
    G = {}
    # fragment offsets of mlp_bwd_s16.hip's LayoutB<KE, KD>
    F_HV, F_FEAT, F_DIRS = 0, 8, 72
    F_H8 = F_DIRS + 8 * KD
    F_L7 = F_H8 + 144
    F_E5 = F_L7 + 3 * 128
    F_L4 = F_E5 + 16 * KE
    F_E0 = F_L4 + 4 * 128
    assert stream.shape[0] >= F_E0 + 16 * KE
    if (LX, LD) == (10, 4):
        assert (F_H8, F_L7, F_E5, F_L4, F_E0) == (80, 224, 608, 640, 1152)

    def enc_slots(f0, npair, x, L, n_cols):
        """FRAG_TE16 products: per point, gradient of every encoding column (from the lane that owns its slot)."""
        out = np.zeros((32, n_cols), np.float64)
        for pr in range(npair):
            acc = w.tile_pair(f0 + pr * 2 * (len(x) // 2), 0, x, len(x) // 2)
            for u in range(2):
                for cc in range(2):
                    for l in range(64):
                        for r in range(4):
                            col = gen16_col(pr, l >> 4, 4 * u + r, L)
                            if col >= 0:
                                out[16 * cc + (l & 15), col] = acc[u][cc][l, r]
        return out

    G["hv"] = tlayer(F_HV, 4, Grgb, 1, mask=keep["hv"])
    G["feat"] = tlayer(F_FEAT, 8, G["hv"], 4)
    g_dirs = enc_slots(F_DIRS, KD, G["hv"], LD, icv)
    g = tlayer(F_H8, 8, G["feat"], 8, Gsig, 1, mask=keep["h"][7])
    G[7] = g
    g_e = None
    for n, l in enumerate(range(6, -1, -1)):
        f0 = F_L7 + 128 * n if l >= 4 else F_L4 + 128 * (3 - l)
        if l == 4:
            g_e = enc_slots(F_E5, KE, g, LX, ic)             # g here is the gradient of pts_linears.5's pre-activation
        g = tlayer(f0, 8, g, 8, mask=keep["h"][l])
        G[l] = g
    g_e = g_e + enc_slots(F_E0, KE, g, LX, ic)

    # plain numpy backward on natural-order rows with the same bf16 roundings
    Wb = {k: bf16_round(v).astype(np.float64) for k, v in sd.items() if k.endswith("weight")}
    rb = lambda v: bf16_round(v.astype(np.float32)).astype(np.float64)   # noqa: E731
    h = [frags_to_rows(f, 8).astype(np.float64) for f in keep["h"]]
    hv = frags_to_rows(keep["hv"], 4).astype(np.float64)
    gr = rb(g_raw)
    g_hv = rb((gr[:, 0:3] @ Wb["rgb_linear.weight"]) * (hv != 0))
    g_feat = rb(g_hv @ Wb["views_linears.0.weight"][:, :256])
    g8 = rb((g_feat @ Wb["feature_linear.weight"] + gr[:, 3:4] @ Wb["alpha_linear.weight"]) * (h[7] != 0))
    ref = {"hv": g_hv, "feat": g_feat, 7: g8}
    g_prev = g8
    for l in range(7, 0, -1):
        Wl = Wb["pts_linears.%d.weight" % l]
        if Wl.shape[1] == ic + 256:
            Wl = Wl[:, ic:]
        g_prev = rb((g_prev @ Wl) * (h[l - 1] != 0))
        ref[l - 1] = g_prev
    for key, n_ks in (("hv", 4), ("feat", 8), (7, 8), (4, 8), (0, 8)):
        got = frags_to_rows(G[key], n_ks)
        np.testing.assert_allclose(got, ref[key], atol=2e-2 * float(np.abs(ref[key]).max()), rtol=0, err_msg=str(key))
    # gradients of the encodings (inputs of the ray gradients)
    ref_dirs = ref["hv"] @ Wb["views_linears.0.weight"][:, 256:256 + icv]
    ref_e = ref[5] @ Wb["pts_linears.5.weight"][:, :ic] + ref[0] @ Wb["pts_linears.0.weight"]
    np.testing.assert_allclose(g_dirs, ref_dirs, atol=2e-2 * float(np.abs(ref_dirs).max()), rtol=0)
    np.testing.assert_allclose(g_e, ref_e, atol=2e-2 * float(np.abs(ref_e).max()), rtol=0)


def test_gen16_layout_is_a_bijection():
    """Every encoding column appears in exactly one (ks, q, j) slot, for every multires in use."""
    for L in (4, 6, 10, 15, 1, 2, 3, 5, 7, 20):
        cols = [gen16_col(ks, q, j, L) for ks in range(gen16_ksteps(L)) for q in range(4) for j in range(8)]
        real = sorted(c for c in cols if c >= 0)
        assert real == list(range(3 + 6 * L)), L


@pytest.mark.parametrize("arch", CASES, ids=["vd_10_4", "vd_15_6", "novd_10"])
def test_stream_matches_kernel_dataflow(arch):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-3, 3, size=(32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    sd = synth.make_state_dict(3, 3.0, **{**arch, "skips": tuple(arch["skips"])})
    emu = emulate(arch, sd, pts, dirs)
    ref = plain_bf16(arch, sd, pts, dirs)
    # identical roundings, different summation order only
    np.testing.assert_allclose(emu, ref, atol=2e-3, rtol=2e-3)
    # and the bf16 network is a faithful approximation of the fp32 oracle
    sdt = O.state_dict_to_torch(sd)
    a = O.Arch(**arch)
    full = O.nerf_forward(sdt, a, torch.from_numpy(pts)[:, None, :],
                          torch.from_numpy(dirs) if arch["use_viewdirs"] else None).reshape(32, -1).numpy()
    assert np.abs(emu - full).max() < 0.35 * max(1.0, np.abs(full).max())


def test_stream_sizes_and_padding():
    arch = CASES[0]
    sd = synth.make_state_dict(3, 1.0, **{**arch, "skips": tuple(arch["skips"])})
    stream, bias = host_pack(arch, sd)
    assert stream.shape[0] % 192 == 0 and stream.shape[0] >= 1184    # padded to whole blocks of any shape
    assert bias.shape[0] == 78
    assert np.all(stream[1184:] == 0)                                 # padding fragments are zero
    # alpha tile: only output row 0 carries weights
    alpha0 = 8 * 4 + 4 * 128 + 8 * 20 + 2 * 128 + 128
    assert np.all(stream[alpha0:alpha0 + 16, 1:32] == 0) and np.all(stream[alpha0:alpha0 + 16, 33:64] == 0)
    assert np.any(stream[alpha0, 0] != 0)


def test_unsupported_arch_reports_error():
    arch = _lib.make_arch(4, 128, 4, [1], True, 6, 2, 0)
    nf = ctypes.c_int64()
    rc = lib.nerf_amd_pack_bf16_host(ctypes.byref(arch), 32, None, None, 0, None, ctypes.byref(nf), None, None)
    assert rc == -3 and b"D=8" in lib.nerf_amd_last_error()


# ---------------------------------------------------------------- split-precision stream (mlp_split.hip, shape 18)
def split_np(x):
    """x -> (hi, lo) as fp32 arrays holding fp16 values: hi = fp16(x) (0 where denormal), lo = fp16((x - hi) 2^11)."""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    hi = np.where(np.abs(x) < 2.0 ** -14, np.float16(0), hi).astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def host_pack_split(arch_kwargs, sd, shape=18):
    names = ["pts_linears.%d" % i for i in range(arch_kwargs["D"])]
    names += ["feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"] if arch_kwargs["use_viewdirs"] else ["output_linear"]
    ws = [np.ascontiguousarray(sd[n + ".weight"], np.float32) for n in names]
    bs = [np.ascontiguousarray(sd[n + ".bias"], np.float32) for n in names]
    arch = _lib.make_arch(arch_kwargs["D"], arch_kwargs["W"], arch_kwargs["output_ch"], arch_kwargs["skips"],
                          arch_kwargs["use_viewdirs"], arch_kwargs["multires"], arch_kwargs["multires_views"], 0)
    n = len(names)
    wp = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    nf, nb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.nerf_amd_pack_bf16_host(ctypes.byref(arch), shape, wp, bp, n, None, ctypes.byref(nf), None, ctypes.byref(nb)), "size query")
    assert nb.value == 0
    stream = np.zeros(nf.value * 512, np.uint16)
    _lib.check(lib.nerf_amd_pack_bf16_host(ctypes.byref(arch), shape, wp, bp, n,
                                           stream.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), ctypes.byref(nf), None, None), "pack")
    return stream.view(np.float16).astype(np.float32).reshape(nf.value, 64, 8)


class WaveEmuSplit(WaveEmu16):
    """One wave = 16 points; activation fragments come as (hi, lo) pairs: x[2k] = hi, x[2k+1] = lo (mlp_split.hip)."""

    def pair(self, f0, t, x1, k1, x2=None, k2=0):
        ah, al = [self.bias_init(t), self.bias_init(t + 1)], [np.zeros((64, 4)), np.zeros((64, 4))]
        n = f0
        for xs, kk in ((x1, k1), (x2, k2)):
            for k in range(kk):
                w0h, w1h, w0l, w1l = self.stream[n], self.stream[n + 1], self.stream[n + 2], self.stream[n + 3]
                n += 4
                xh, xl = xs[2 * k], xs[2 * k + 1]
                ah[0] = self.mfma(w0h, xh, ah[0]); ah[1] = self.mfma(w1h, xh, ah[1])
                al[0] = self.mfma(w0h, xl, al[0]); al[1] = self.mfma(w1h, xl, al[1])
                al[0] = self.mfma(w0l, xh, al[0]); al[1] = self.mfma(w1l, xh, al[1])
        return [(ah[u] + al[u] / 2048.0).astype(np.float32) for u in range(2)]

    def single(self, f0, t, x1, k1):
        ah, al = self.bias_init(t), np.zeros((64, 4))
        for k in range(k1):
            wh, wl = self.stream[f0 + 2 * k], self.stream[f0 + 2 * k + 1]
            ah = self.mfma(wh, x1[2 * k], ah)
            al = self.mfma(wh, x1[2 * k + 1], al)
            al = self.mfma(wl, x1[2 * k], al)
        return (ah + al / 2048.0).astype(np.float32)

    def layer(self, f0, t0, npair, x1, k1, x2=None, k2=0, relu=True):
        y = []
        for p in range(npair):
            ev, od = self.pair(f0 + p * 4 * (k1 + k2), t0 + 2 * p, x1, k1, x2, k2)
            v = np.concatenate([ev, od], -1)
            if relu:
                v = np.maximum(v, 0)
            y += list(split_np(v))
        return y

    @staticmethod
    def encode(emb, L, K):
        out = []
        for ks in range(K):
            f = np.zeros((64, 8), np.float32)
            for l in range(64):
                for j in range(8):
                    c = gen16_col(ks, l >> 4, j, L)
                    f[l, j] = emb[l & 15, c] if c >= 0 else 0.0
            out += list(split_np(f))
        return out


def emulate_split(arch, sd, pts, dirs):
    LX, LD, VD = arch["multires"], arch["multires_views"], arch["use_viewdirs"]
    KE, KD = gen16_ksteps(LX), (gen16_ksteps(LD) if VD else 0)
    _, bias = host_pack(arch, sd, 16)
    stream = host_pack_split(arch, sd)
    F_L1 = 32 * KE
    F_L5 = F_L1 + 4 * 256
    F_L6 = F_L5 + 32 * (KE + 8)
    F_HEAD = F_L6 + 512
    w = WaveEmuSplit(stream, bias)
    E = w.encode(O.embed(torch.from_numpy(pts), LX).numpy(), LX, KE)
    A = w.layer(0, 0, 8, E, KE)
    B = w.layer(F_L1, 16, 8, A, 8)
    A = w.layer(F_L1 + 256, 32, 8, B, 8)
    B = w.layer(F_L1 + 512, 48, 8, A, 8)
    A = w.layer(F_L1 + 768, 64, 8, B, 8)
    B = w.layer(F_L5, 80, 8, E, KE, A, 8)
    A = w.layer(F_L6, 96, 8, B, 8)
    B = w.layer(F_L6 + 256, 112, 8, A, 8)
    if VD:
        Dv = w.encode(O.embed(torch.from_numpy(dirs), LD).numpy(), LD, KD)
        F_ALPHA = F_HEAD + 256
        F_VIEWS = F_ALPHA + 16
        F_RGB = F_VIEWS + 16 * (8 + KD)
        assert F_RGB + 8 <= stream.shape[0]
        A = w.layer(F_HEAD, 128, 8, B, 8, relu=False)
        alpha = w.single(F_ALPHA, 144, B, 8)
        B2 = w.layer(F_VIEWS, 145, 4, A, 8, Dv, KD)
        rgb = w.single(F_RGB, 153, B2, 4)
        return np.concatenate([rgb[:16, 0:3], alpha[:16, 0:1]], -1)
    o = w.single(F_HEAD, 128, B, 8)
    out = np.zeros((16, arch["output_ch"]), np.float32)
    for qq in range(4):
        for r in range(4):
            if 4 * qq + r < arch["output_ch"]:
                out[:, 4 * qq + r] = o[16 * qq:16 * qq + 16, r]
    return out


def test_split_stream_values_are_exact_fp16_pairs():
    """pack.hip's hand-written fp32 -> fp16 conversion against numpy's, and the pair's accuracy: hi + lo / 2^11
    reproduces every weight to 2^-21 relative (22 significant bits), no fragment value is an fp16 denormal hi part."""
    arch = CASES[0]
    sd = synth.make_state_dict(3, 3.0, **{**arch, "skips": tuple(arch["skips"])})
    # make the conversion meet denormals, ties and large values as well
    w0 = sd["pts_linears.1.weight"]
    w0[0, :8] = [1e-6, 6.0e-5, 6.2e-5, 3.0e-8, 1000.0, 65504.0, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11]
    stream = host_pack_split(arch, sd)
    assert stream.shape[0] % 192 == 0 and stream.shape[0] >= 2344
    # fragment order inside a pair and k-step: hi(t0), hi(t1), lo(t0), lo(t1); layer 1 = pts_linears.1 starts at 32 * KE
    f0 = 32 * gen16_ksteps(10)
    W = sd["pts_linears.1.weight"]
    for u in range(2):
        hi, lo = stream[f0 + u], stream[f0 + 2 + u]
        for l in range(64):
            for j in range(8):
                col = 32 * 0 + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3)          # acc16_col(ks = 0, q, j)
                wv = np.float32(W[16 * u + (l & 15), col])
                eh, el = split_np(wv)
                assert hi[l, j] == eh and lo[l, j] == el, (u, l, j, wv, hi[l, j], eh, lo[l, j], el)
                assert abs(np.float64(hi[l, j]) + np.float64(lo[l, j]) / 2048.0 - np.float64(wv)) <= 2.0 ** -21 * abs(np.float64(wv)) + 2.0 ** -25   # below 2^-14 the lo part alone carries the value


@pytest.mark.parametrize("arch", CASES, ids=["vd_10_4", "vd_15_6", "novd_10"])
def test_split_stream_matches_kernel_dataflow_at_fp32_accuracy(arch):
    """mlp_split.hip's data flow replayed on the host-packed fp16 (hi, lo) stream: against an fp64 evaluation of
    the fp32 weights it is inside the fp32 parity gate (1e-4 + 1e-4 |y|) on the x3 weights -- where bf16 pairs
    are not."""
    rng = np.random.default_rng(7)
    pts = rng.uniform(-3, 3, size=(16, 3)).astype(np.float32)
    dirs = rng.normal(size=(16, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    sd = synth.make_state_dict(3, 3.0, **{**arch, "skips": tuple(arch["skips"])})
    got = emulate_split(arch, sd, pts, dirs)
    sd64 = {k: torch.from_numpy(np.asarray(v)).double() for k, v in sd.items()}
    a = O.Arch(**{**arch, "skips": tuple(arch["skips"])})
    e = O.embed(torch.from_numpy(pts), arch["multires"])
    if arch["use_viewdirs"]:
        e = torch.cat([e, O.embed(torch.from_numpy(dirs), arch["multires_views"])], -1)
    want = O.mlp(sd64, a, e.double()).numpy()
    err = np.abs(got - want)
    assert np.abs(want).max() > 1.0
    assert (err <= 0.5 * (1e-4 + 1e-4 * np.abs(want))).all(), (err.max(), (err / (1e-4 + 1e-4 * np.abs(want))).max())


@pytest.mark.parametrize("arch", CASES, ids=["vd_10_4", "vd_15_6", "novd_10"])
def test_backward_split_stream_is_the_backward_stream_as_fp16_pairs(arch):
    """The transposed stream of the split-precision dX chain (shape 19, mlp_bwd_split.hip) holds the same weights in the same
    fragment order as the bf16 backward stream (shape 17, replayed by test_backward_stream_matches_kernel_dataflow): per
    pair of tiles and k-step the two bf16 fragments t0, t1 become hi(t0), hi(t1), lo(t0), lo(t1), and hi + lo / 2^11 is the
    fp32 weight the bf16 fragment rounds (to 22 bits; the same zero pattern)."""
    sd = synth.make_state_dict(3, 2.0, **{**arch, "skips": tuple(arch["skips"])})
    b16, _ = host_pack(arch, sd, 17)
    sp = host_pack_split(arch, sd, 19)
    n_used = {(10, 4, True): 1184, (15, 6, True): 1224, (10, 4, False): 976}[(arch["multires"], arch["multires_views"], arch["use_viewdirs"])]
    assert b16.shape[0] % 192 == 0 and sp.shape[0] % 192 == 0 and sp.shape[0] >= 2 * n_used
    assert not np.any(sp[2 * n_used:]) and not np.any(b16[n_used:])
    for m in range(n_used // 2):
        for u in range(2):
            hi, lo, want = sp[4 * m + u].astype(np.float64), sp[4 * m + 2 + u].astype(np.float64), b16[2 * m + u].astype(np.float64)
            val = hi + lo / 2048.0
            assert np.array_equal(val == 0, want == 0) or np.all(np.abs(val[(val == 0) != (want == 0)]) < 1e-38), m
            # bf16 keeps 8 bits: the pair's value rounds to the bf16 fragment (or its neighbour on a tie of the last pair bit)
            assert np.all(np.abs(val - want) <= 2.0 ** -8 * np.abs(val) + 1e-30), (m, u, float(np.abs(val - want).max()))
    # the pair itself is exact to 2^-21 against the fp32 weight: pts_linears.7's transposed fragment 0 of the g_h(6) layer
    W = np.asarray(sd["pts_linears.7.weight"], np.float32)
    KD = gen16_ksteps(arch["multires_views"]) if arch["use_viewdirs"] else 1
    f_l7 = (144 + 16 * KD + 8 * 36) if arch["use_viewdirs"] else 8 * 4           # LayoutBS::F_L7
    hi, lo = sp[f_l7].astype(np.float64), sp[f_l7 + 2].astype(np.float64)
    for l in range(64):
        for j in range(8):
            o = 16 * (j >> 2) + 4 * (l >> 4) + (j & 3)                             # acc16_col(ks = 0, q, j): output feature
            wv = np.float64(W[o, l & 15])                                          # element = W[o][input feature row0 + (l & 15)]
            assert abs(hi[l, j] + lo[l, j] / 2048.0 - wv) <= 2.0 ** -21 * abs(wv) + 2.0 ** -25, (l, j)
