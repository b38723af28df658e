"""Host side of the convolution engine (csrc/conv.hip): weight packing with the eval-mode
batch norm folded in, and tensor wrappers over bmv_conv_fwd / bmv_conv3d_transpose_fwd /
bmv_fpn_topdown_fwd.

The conv blocks of the reference (`ConvBnReLU`, `ConvBnReLU3D`, lib/networks/enerf/utils.py:10-33)
become one launch each; their parameters stay ordinary nn.Module parameters with the reference's
state-dict names (networks/enerf/cnn.py) and are re-packed only when one of them changes.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib, ktimer, ops, switches
from ._lib import dptr, stream


def fold_bn(weight, bn, out_dim=0):
    """weight of a conv (out_dim=0) or transposed conv (out_dim=1) followed by eval-mode batch norm
    -> (weight * gamma / sqrt(var + eps), beta - mean * gamma / sqrt(var + eps))."""
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
    shift = bn.bias.detach() - bn.running_mean.detach() * scale
    shape = [1] * weight.dim()
    shape[out_dim] = -1
    return weight.detach() * scale.view(shape), shift


def pack_conv(weight, bias, stride=1, allow_pair=True):
    """weight (Cout,Cin,[kd,]kh,kw), bias (Cout) or None -> (wpack, bias16) in the layout of include/bmv.h:
    wpack[t][c][tap][k][o] = weight[16 t + o][4 c + k][tap], or the row-pair layout when the library pairs
    output rows for this shape (Cout <= 8, 3x3 / 3x3x3, stride 1)."""
    Cout, Cin = weight.shape[:2]
    k = weight.shape[-1]
    kd = weight.shape[2] if weight.dim() == 5 else 1
    taps = int(weight[0, 0].numel())
    nt, nc = (Cout + 15) // 16, (Cin + 3) // 4
    dev = weight.device
    b = torch.zeros(nt * 16, device=dev, dtype=torch.float32)
    if bias is not None:
        b[:Cout] = bias.detach().float()
    if allow_pair and _lib.load().bmv_conv_pairs_rows(Cout, kd, k, stride):
        w8 = torch.zeros(8, nc * 4, kd, k, k, device=dev, dtype=torch.float32)
        w8[:Cout, :Cin] = weight.detach().reshape(Cout, Cin, kd, k, k).float()
        w8 = w8.view(8, nc, 4, kd, k, k).permute(1, 3, 4, 5, 2, 0)          # (nc, kd, ky, kx, 4, 8)
        wp = torch.zeros(nc, kd, k + 1, k, 4, 16, device=dev, dtype=torch.float32)
        wp[:, :, :k, :, :, :8] = w8                                         # output row y   meets filter row j
        wp[:, :, 1:, :, :, 8:] = w8                                         # output row y+1 meets filter row j-1
        return wp.view(1, nc, kd * (k + 1) * k, 4, 16).contiguous(), b
    w = torch.zeros(nt * 16, nc * 4, taps, device=dev, dtype=torch.float32)
    w[:Cout, :Cin] = weight.detach().reshape(Cout, Cin, taps).float()
    wpack = w.view(nt, 16, nc, 4, taps).permute(0, 2, 4, 3, 1).contiguous()
    return wpack, b


def pack_conv_c4(weight, bias):
    """weight (Cout <= 12, Cin, [3,] 3, 3), bias (Cout) or None -> (wpack, bias) of bmv_conv_c4_fwd (csrc/conv_c4.hip):
    wpack[chunk][tap][g][i][k] = weight[4 g + i][4 chunk + k][tap], zero padded."""
    Cout, Cin = weight.shape[:2]
    taps = int(weight[0, 0].numel())
    ng, nc = (Cout + 3) // 4, (Cin + 3) // 4
    dev = weight.device
    w = torch.zeros(ng * 4, nc * 4, taps, device=dev, dtype=torch.float32)
    w[:Cout, :Cin] = weight.detach().reshape(Cout, Cin, taps).float()
    wpack = w.view(ng, 4, nc, 4, taps).permute(2, 4, 0, 1, 3).contiguous()       # (chunk, tap, g, i, k)
    b = torch.zeros(ng * 4, device=dev, dtype=torch.float32)
    if bias is not None:
        b[:Cout] = bias.detach().float()
    return wpack, b


def conv_c4_fwd(x, wpack, bias, Cout, relu=False, slope=None, records=False, variant=0, quad_out=False):
    """x (B,Cin,D,H,W) or (B,Cin,H,W) -> act(conv(x, 3x3[x3], stride 1, padding 1) + bias) with Cout <= 12 output
    channels on the 4-row matrix blocks (csrc/conv_c4.hip).  records: the renderer's volume records (VolumeRecords of
    channels 0..7 + the planar channel 8 when Cout == 9) instead of the planar tensor.  x may be an ops.QuadVolume (the
    plane sweep's quad-record output, (B,Cin/4,D,H,W,4)): staged with 16-byte loads (include/bmv.h: mode | 4).
    quad_out (3-D, Cout % 4 == 0): the result as an ops.QuadVolume (mode 8) for a consumer that stages 16-byte records."""
    qin = isinstance(x, ops.QuadVolume)
    if qin:
        x = x.data
        B, Q, D, H, W, _ = x.shape
        Cin, is3d = 4 * Q, True
    elif x.dim() == 5:
        is3d = True
        B, Cin, D, H, W = x.shape
    else:
        is3d = False
        B, Cin, H, W = x.shape
        D = 1
    kd = 3 if is3d else 1
    lib = _lib.load()
    assert wpack.numel() == lib.bmv_conv_c4_wpack_floats(Cout, Cin, kd)
    x = x if x.is_contiguous() else x.contiguous()
    out2 = None
    quad_out = bool(quad_out) and is3d and Cout % 4 == 0 and not records
    if records:
        out = torch.empty(B, D, H, W, 8, device=x.device, dtype=torch.float32)
        out2 = torch.empty(B, D, H, W, device=x.device, dtype=torch.float32) if Cout == 9 else None
    elif quad_out:
        out = torch.empty(B, Cout // 4, D, H, W, 4, device=x.device, dtype=torch.float32)
    else:
        out = torch.empty((B, Cout, D, H, W) if is3d else (B, Cout, H, W), device=x.device, dtype=torch.float32)
    with ktimer.region(f"conv_c4[{Cin}->{Cout},{D}x{H}x{W}]"):
        rc = lib.bmv_conv_c4_fwd(dptr(x, "conv input"), dptr(wpack, "wpack"), dptr(bias, "bias"), dptr(out),
                                 dptr(out2) if out2 is not None else None, B, Cin, D, H, W, Cout, kd, _slope(relu, slope),
                                 (2 if records else 8 if quad_out else 0) | (4 if qin else 0), int(variant), stream())
    _lib.check(rc, "conv_c4_fwd")
    if records:
        return VolumeRecords(out), out2
    return ops.QuadVolume(out) if quad_out else out


def pack_convT_c4(weight, bias):
    """ConvTranspose3d weight (Cin, Cout <= 8, 3, 3, 3) (batch norm folded: fold_bn(..., out_dim=1)), bias (Cout) ->
    (wpack, bias) of bmv_conv3d_transpose_c4_fwd: wpack[chunk][tap][g][i][k] = weight[4 chunk + k][4 g + i][tap]."""
    return pack_conv_c4(weight.detach().transpose(0, 1), bias)


def convT_c4_fwd(x, wpack, bias, Cout, skip=None, relu=False, slope=None, variant=0, quad_out=False):
    """x (B,Cin,D,H,W) -> act(conv_transpose3d(x, k=3, stride=2, padding=1, output_padding=1) + bias) + skip, Cout <= 8,
    on the 4 x 4 x 1 matrix blocks (csrc/conv_c4.hip).  quad_out (Cout = 8): the result as an ops.QuadVolume (quad
    records, for conv_c4_fwd's 16-byte staging); `skip` is planar either way."""
    B, Cin, D, H, W = x.shape
    quad_out = bool(quad_out) and Cout == 8 and variant == 0
    if quad_out:
        out = torch.empty(B, 2, 2 * D, 2 * H, 2 * W, 4, device=x.device, dtype=torch.float32)
    else:
        out = torch.empty(B, Cout, 2 * D, 2 * H, 2 * W, device=x.device, dtype=torch.float32)
    skip_quad = isinstance(skip, ops.QuadVolume)
    if skip_quad and not quad_out:
        skip, skip_quad = skip.to_planar().contiguous(), False
    if skip is not None:
        assert tuple(skip.shape) == (B, Cout, 2 * D, 2 * H, 2 * W)
        skip = skip.data if skip_quad else skip
        assert skip.is_contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    lib = _lib.load()
    with ktimer.region(f"convT_c4[{Cin}->{Cout},{D}x{H}x{W}]"):
        rc = lib.bmv_conv3d_transpose_c4_fwd(dptr(x, "conv input"), dptr(wpack, "wpack"), dptr(bias, "bias"),
                                  dptr(skip) if skip is not None else None, dptr(out), B, Cin, D, H, W, Cout,
                                  _slope(relu, slope), int(variant) | (16 if quad_out else 0) | (32 if skip_quad else 0), stream())
    _lib.check(rc, "convT_c4_fwd")
    return ops.QuadVolume(out) if quad_out else out


def pack_conv_split(weight, bias, parts=3):
    """weight (Cout <= 16, Cin % 8 == 0, 3, 3, 3) -> the split-bf16 A operands of bmv_conv3d_split_fwd (csrc/conv_split.hip),
    int32 [octet][step 7][part][lane 64][4]: lane = 16 * (tap % 4) + cout, the 8 bf16 of a lane = the 8 channels of the
    octet at tap 4 * step + lane // 16.  parts = 3: hi / mid / lo = the three 8-bit pieces of the fp32 mantissa (exact);
    parts = 2: hi = upper 16 bits, lo = bf16(w - hi) rounded to nearest even.  Returns (wsplit, bias16, parts)."""
    Cout, Cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3, 3) and Cin % 8 == 0 and Cout <= 16 and parts in (2, 3)
    dev = weight.device
    w = torch.zeros(16, Cin, 28, device=dev, dtype=torch.float32)
    w[:Cout, :, :27] = weight.detach().reshape(Cout, Cin, 27).float()
    w = w.view(16, Cin // 8, 8, 7, 4).permute(1, 3, 4, 0, 2).reshape(Cin // 8, 7, 64, 8).contiguous()   # (o, g, kk*16+m, c)
    trunc = lambda t: (t.view(torch.int32) & -65536).view(torch.float32)
    top16 = lambda t: (t.view(torch.int32) >> 16) & 0xFFFF
    rne16 = lambda t: t.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    pieces, r = [], w
    for i in range(parts):
        if i + 1 < parts:
            h = trunc(r)
            pieces.append(top16(h))
            r = r - h
        else:
            pieces.append(rne16(r))
    pack = lambda t: t[..., 0::2] | (t[..., 1::2] << 16)
    wsplit = torch.stack([pack(t) for t in pieces], 2).contiguous()                                     # (o, g, P, 64, 4)
    b = torch.zeros(16, device=dev, dtype=torch.float32)
    if bias is not None:
        b[:Cout] = bias.detach().float()
    return wsplit, b, parts


def conv3d_split_fwd(x, wsplit, bias, parts, Cout, relu=False, slope=None, out=None):
    """x (B,Cin,D,H,W) -> act(conv3d(x, k=3, padding=1) + bias) on the split-bf16 path (see pack_conv_split)."""
    B, Cin, D, H, W = x.shape
    lib = _lib.load()
    assert wsplit.numel() == lib.bmv_conv3d_split_wsplit_ints(Cin, parts) and wsplit.dtype == torch.int32
    if out is None:
        out = torch.empty(B, Cout, D, H, W, device=x.device, dtype=torch.float32)
    x = x if x.is_contiguous() else x.contiguous()
    with ktimer.region(f"conv_split{parts}[{Cin}->{Cout},{D}x{H}x{W}]"):
        rc = lib.bmv_conv3d_split_fwd(dptr(x, "conv input"), dptr(wsplit, "wsplit", torch.int32), parts, dptr(bias, "bias"),
                                      dptr(out), B, Cin, D, H, W, Cout, _slope(relu, slope), stream())
    _lib.check(rc, "conv3d_split_fwd")
    return out


def _split3_words(w):
    """float32 tensor (..., 8) -> three int32 tensors (..., 4): the hi / mid / lo bf16 pieces of every value (hi = the upper
    16 bits, mid = the upper 16 bits of w - hi, lo = w - hi - mid, which has at most 8 significant bits left: hi + mid + lo
    == w EXACTLY), packed in pairs [even | odd << 16] -- the arithmetic of csrc/conv_c4s.hip's staging, on the host."""
    trunc = lambda t: (t.view(torch.int32) & -65536).view(torch.float32)      # noqa: E731
    top16 = lambda t: (t.view(torch.int32) >> 16) & 0xFFFF                       # noqa: E731
    hi = trunc(w)
    r = w - hi
    mid = trunc(r)
    lo = (r - mid).to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
    pack = lambda t: t[..., 0::2] | (t[..., 1::2] << 16)                         # noqa: E731
    return pack(top16(hi)), pack(top16(mid)), pack(lo)


def pack_conv_c4s(weight, bias, pair=None):
    """weight (Cout, Cin % 8 == 0, 3, 3, 3), bias (Cout) or None -> (wsplit, bias16, pair) of bmv_conv_c4s_fwd
    (csrc/conv_c4s.hip; include/bmv.h has the layout): int32 [octet][step 3][kz 3][piece 3][lane 64][4]; lane = 16 kk + m
    holds, as 8 bf16 per piece, the octet's 8 input channels of matrix row m at in-plane slot t = 4 step + kk.  pair
    (default: Cout == 8): the row-paired form -- row m = (output row m // 8, channel m % 8), slot t = (j, kx) = (t // 3,
    t % 3) over the 4 input rows j a row pair touches -- else row m = channel m, slots t < 9 = (ky, kx), 9..11 zero."""
    Cout, Cin = weight.shape[:2]
    assert weight.shape[2:] == (3, 3, 3) and Cin % 8 == 0 and Cout <= 16
    pair = (Cout == 8) if pair is None else bool(pair)
    assert not pair or Cout == 8
    dev = weight.device
    w5 = weight.detach().float()
    w = torch.zeros(16, Cin, 3, 12, device=dev, dtype=torch.float32)                # (m, cin, kz, slot)
    if pair:
        wp = torch.zeros(2, 8, Cin, 3, 4, 3, device=dev, dtype=torch.float32)      # (r, c, cin, kz, j, kx)
        wp[0, :, :, :, 0:3] = w5
        wp[1, :, :, :, 1:4] = w5
        w[:] = wp.reshape(16, Cin, 3, 12)
    else:
        w[:Cout, :, :, :9] = w5.reshape(Cout, Cin, 3, 9)
    # (m, o, c, kz, g, kk) -> (o, g, kz, kk, m, c): the order the kernel uses (and therefore loads) them in
    w = w.view(16, Cin // 8, 8, 3, 3, 4).permute(1, 4, 3, 5, 0, 2).reshape(Cin // 8, 3, 3, 64, 8).contiguous()
    wsplit = torch.stack(_split3_words(w), 3).contiguous()                          # (o, g, kz, 3, 64, 4)
    b = torch.zeros(16, device=dev, dtype=torch.float32)
    if bias is not None:
        b[:Cout] = bias.detach().float()
    return wsplit, b, pair


def conv_c4s_fwd(x, wsplit, bias, pair, Cout, relu=False, slope=None, records=False, quad_out=False, split_heads=False):
    """x = ops.QuadVolume (B,Cin/4,D,H,W,4), Cin % 8 == 0 -> act(conv3d(x, 3x3x3, stride 1, padding 1) + bias) on the bf16
    matrix cores with three-piece fp32 operands (csrc/conv_c4s.hip: fp32 accuracy).  records: the renderer's volume
    records (VolumeRecords of channels 0..7 + the planar channel 8 when Cout == 9); quad_out (Cout % 4 == 0): the result
    as an ops.QuadVolume; else the planar tensor (split_heads with Cout == 9: the pair (channels 0..7, channel 8) as
    strided views of the kernel's two outputs instead of one concatenated copy)."""
    assert isinstance(x, ops.QuadVolume), "conv_c4s_fwd stages quad records (ops.QuadVolume)"
    xd = x.data
    B, Q, D, H, W, _ = xd.shape
    Cin = 4 * Q
    lib = _lib.load()
    assert wsplit.dtype == torch.int32 and wsplit.numel() == lib.bmv_conv_c4s_wsplit_ints(Cin, int(pair))
    xd = xd if xd.is_contiguous() else xd.contiguous()
    out2 = None
    want_planar = not records and not (quad_out and Cout % 4 == 0)
    # the kernel writes 16-byte records only: a planar result is a strided VIEW of quad records (Cout % 4 == 0) or of the
    # volume records + the planar ninth channel (Cout == 9), made contiguous where a caller needs that
    as_records = records or (want_planar and Cout == 9)
    if as_records:
        assert Cout in (8, 9)
        out = torch.empty(B, D, H, W, 8, device=xd.device, dtype=torch.float32)
        if Cout == 9:
            out2 = torch.empty(B, D, H, W, device=xd.device, dtype=torch.float32)
    else:
        assert Cout % 4 == 0, "conv_c4s_fwd: planar / quad output needs Cout % 4 == 0 (or the 8 + 1 heads)"
        out = torch.empty(B, Cout // 4, D, H, W, 4, device=xd.device, dtype=torch.float32)
    with ktimer.region(f"conv_c4s[{Cin}->{Cout},{D}x{H}x{W}]"):
        rc = lib.bmv_conv_c4s_fwd(dptr(xd, "conv input"), dptr(wsplit, "wsplit", torch.int32), dptr(bias, "bias"), dptr(out),
                                  dptr(out2) if out2 is not None else None, B, Cin, D, H, W, Cout, int(pair),
                                  _slope(relu, slope), 2 if as_records else 8, stream())
    _lib.check(rc, "conv_c4s_fwd")
    if records:
        return VolumeRecords(out), out2
    if want_planar:
        if Cout == 9:
            if split_heads:      # (feat_conv, depth_conv) as the reference returns them: views, no copy
                return out.permute(0, 4, 1, 2, 3), out2
            return torch.cat([out.permute(0, 4, 1, 2, 3), out2[:, None]], 1)
        return ops.QuadVolume(out).to_planar()
    return ops.QuadVolume(out)


def conv3d_split_heads_records(x, wsplit, bias, parts):
    """conv_heads_records on the split-bf16 path (weights packed with pack_conv_split in VolumeRecords.ORDER)."""
    B, Cin, D, H, W = x.shape
    rec = torch.empty(B, D, H, W, 8, device=x.device, dtype=torch.float32)
    dp = torch.empty(B, D, H, W, device=x.device, dtype=torch.float32)
    x = x if x.is_contiguous() else x.contiguous()
    lib = _lib.load()
    with ktimer.region(f"conv_split{parts}[{Cin}->9 records,{D}x{H}x{W}]"):
        rc = lib.bmv_conv3d_split_heads_fwd(dptr(x, "conv input"), dptr(wsplit, "wsplit", torch.int32), parts, dptr(bias, "bias"),
                                            dptr(rec), dptr(dp), B, Cin, D, H, W, stream())
    _lib.check(rc, "conv3d_split_heads_fwd")
    return VolumeRecords(rec), dp


# Which of the regularisers' first layers / heads run on the split-bf16 path (csrc/conv_split.hip); BMV_CONV_SPLIT:
#   "0" (default): fp32 MFMA engine everywhere;
#   "auto": THREE bf16 pieces per operand (fp32-equivalent: the operand exactly, six MFMAs per product group) for the
#           heads and for a first layer with >= 32 input channels (stand-alone 77 -> 61, 52 -> 47, 30 -> 23 us; in the
#           frame only -7 us of 881: the level-0 first layer runs beside FeatureNet's top-down path and is no faster
#           there -- all 183 GPU tests pass unchanged with it, but it is not worth a default);
#   "3" / "2": all four layers with three / two pieces (two = 2^-16 per product: the opt-in experiment, +7 % of the frame).
def split_bf16_default():
    """The policy a regulariser module is constructed with (`_CostReg.split_bf16`): read from the switch WHEN the module
    is built, so `switches.set("BMV_CONV_SPLIT", ...)` after import counts (module constants froze it)."""
    raw = str(switches.get("BMV_CONV_SPLIT")).strip()
    return raw if raw == "auto" else int(raw)


# The regularisers' first layers and heads on v_mfma_f32_4x4x1_16b_f32 (csrc/conv_c4.hip): BMV_CONV_C4=1 / 0, read by the
# modules at construction (`switches.on("BMV_CONV_C4")`)


def split_parts(policy, kind, cin):
    """pieces to use for a layer (0 = fp32 engine); kind: "conv0" | "heads"."""
    if policy == "auto":
        return 3 if (kind == "heads" or cin >= 32) else 0
    return int(policy)


_ZERO_BIAS = {}


def _zero_bias(n, device):
    key = (n, str(device))
    if key not in _ZERO_BIAS:
        _ZERO_BIAS[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return _ZERO_BIAS[key]


_PACK_DEV_CACHE = {}


def clear_pack_cache():
    """Drop the device-side weight packs of pack_conv_dev (train.GraphedTrainStep calls this around a capture: a pack made
    while capturing is an allocation of that graph's private pool)."""
    _PACK_DEV_CACHE.clear()


def pack_conv_dev(weight, bias, stride=1, transposed=False, flip=False, for_convT=False):
    """pack_conv / pack_convT in ONE launch (bmv_conv_pack_weights), for weights that change every step.
    `weight` is torch's tensor as stored; the blob is for a convolution whose (Cout, Cin) are weight.shape[:2], or
    shape[1::-1] when `transposed`; `flip` reverses the taps (transposed + flip: the data gradient of a stride-1
    convolution as a convolution); `for_convT`: the blob feeds convT3d_fwd.  Returns (wpack, bias16); without a bias
    the (cached) zero vector."""
    lib = _lib.load()
    # One pack per (parameter version, form) and step: the K cost volumes of a boost step run the same regulariser K
    # times (and forward + the data gradient of a layer ask for different forms): config 5 launched 173 pack kernels
    # per step.  Keyed by storage + version: an optimiser step bumps the version and the entry is replaced
    # (a graph REPLAY updates parameters without touching the counter: train.GraphedTrainStep clears the cache after one).
    key = (weight.data_ptr(), tuple(weight.shape), stride, bool(transposed), bool(flip), bool(for_convT),
           None if bias is None else bias.data_ptr(), torch.cuda.is_current_stream_capturing() if weight.is_cuda else False)
    ver = (weight._version, None if bias is None else bias._version)
    hit = _PACK_DEV_CACHE.get(key)
    # (the entry remembers the tensor OBJECTS, weakly: a temporary -- the parity sub-filters of conv_train's 5x5 data
    # gradient -- or the parameter of a network that went away leaves a dead reference, and a new tensor that the caching
    # allocator puts at the same address with the same version counter is not taken for it)
    if hit is not None and hit[0] == ver and hit[3]() is weight and (bias is None or hit[4]() is bias):
        return hit[1], hit[2]
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    Cout, Cin = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    k = w.shape[-1]
    kd = w.shape[2] if w.dim() == 5 else 1
    n = ((Cout + 15) // 16) * ((Cin + 3) // 4) * kd * k * k * 64 if for_convT else lib.bmv_conv_wpack_floats(Cin, Cout, kd, k, stride)
    wpack = torch.empty(n, device=w.device, dtype=torch.float32)
    _lib.check(lib.bmv_conv_pack_weights(dptr(w, "weight"), Cin, Cout, kd, k, stride, int(transposed), int(flip), int(for_convT),
                                         dptr(wpack), stream()), "conv_pack_weights")
    nt16 = (Cout + 15) // 16 * 16
    if bias is None:
        b = _zero_bias(nt16, w.device)
    else:
        b = torch.zeros(nt16, device=w.device, dtype=torch.float32)
        b[:Cout] = bias.detach()
    if len(_PACK_DEV_CACHE) > 512:       # (parameters that went away: storage addresses are not reused as keys for ever)
        _PACK_DEV_CACHE.clear()
    _PACK_DEV_CACHE[key] = (ver, wpack, b, weakref.ref(weight), None if bias is None else weakref.ref(bias))
    return wpack, b


def pack_convT(weight, bias):
    """ConvTranspose3d weight (Cin,Cout,3,3,3) -> the same blob layout with the roles of dims 0/1 swapped."""
    return pack_conv(weight.detach().transpose(0, 1), bias, allow_pair=False)


def _slope(relu, slope):
    """activation v > 0 ? v : s * v of the C ABI: relu=True -> 0, slope given -> leaky ReLU, neither -> 1 (identity)."""
    return float(slope) if slope is not None else (0.0 if relu else 1.0)


def convT3d_fwd(x, wpack, bias, Cout, relu=False, skip=None, out=None, slope=None):
    """x (B,Cin,D,H,W) -> act(conv_transpose3d(x, k=3, stride=2, padding=1, output_padding=1) + bias) + skip."""
    B, Cin, D, H, W = x.shape
    shape = (B, Cout, 2 * D, 2 * H, 2 * W)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    assert out.shape == shape and out.is_contiguous()
    if skip is not None:
        assert skip.shape == shape and skip.is_contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    lib = _lib.load()
    with ktimer.region(f"convT3d[{Cin}->{Cout},{D}x{H}x{W}]"):
        rc = lib.bmv_conv3d_transpose_fwd(dptr(x, "convT input"), dptr(wpack, "wpack"), dptr(bias, "bias"),
                                 dptr(skip) if skip is not None else None, dptr(out), B, Cin, D, H, W, Cout,
                                 _slope(relu, slope), stream())
    _lib.check(rc, "convT3d_fwd")
    return out


def fpn_topdown(fine, coarse, weight, bias, out=None):
    """bilinear_x2(coarse, align_corners=True) + conv1x1(fine, weight (C,Cf,1,1), bias) -> (B,C,H,W)."""
    B, Cf, H, W = fine.shape
    from .ops import QuadFeats
    if isinstance(coarse, QuadFeats):          # quad-planar (B,C/4,h,w,4): the plane sweep's layout, read in place
        C = coarse.shape[-3]
        assert tuple(coarse.data.shape) == (B, C // 4, H // 2, W // 2, 4)
        coarse_mem, cl = coarse.data, 3
    else:
        C = coarse.shape[1]
        assert coarse.shape == (B, C, H // 2, W // 2)
        # a (B,C,h,w) view of a channel-last (B,h,w,C) buffer is read in place
        cl = int((not coarse.is_contiguous()) and coarse.permute(0, 2, 3, 1).is_contiguous())
        coarse_mem = coarse.permute(0, 2, 3, 1) if cl else coarse.contiguous()
    if out is None:
        out = torch.empty(B, C, H, W, device=fine.device, dtype=torch.float32)
    w = weight.detach().reshape(C, Cf).contiguous()
    lib = _lib.load()
    with ktimer.region(f"fpn_topdown[{Cf}->{C},{H}x{W}]"):
        rc = lib.bmv_fpn_topdown_fwd(dptr(fine.contiguous(), "fine"), dptr(coarse_mem, "coarse"), dptr(w, "w"),
                                     dptr(bias.detach().contiguous(), "bias"), dptr(out), B, Cf, C, H, W, int(cl), stream())
    _lib.check(rc, "fpn_topdown_fwd")
    return out


class LookupRecords:
    """(..., H, W, 12) per-pixel lookup records of the fused renderer for one or more source views:
    [feature ch 0 2 4 6 | ch 1 3 5 7 | r b | g 0] (include/bmv.h, bmv_render_args.im_packed).  Stands where the planar
    8-channel feature map would (`feats['level_2']`); the source colours ride along."""
    EVEN_ODD = (0, 2, 4, 6, 1, 3, 5, 7)      # output-channel order the smoothing conv's weights are packed in

    def __init__(self, t):
        assert t.shape[-1] == 12
        self.t = t

    def reshape_views(self, B, V):
        return LookupRecords(self.t.reshape(B, V, *self.t.shape[-3:]))

    def __getitem__(self, idx):          # views picked out of an all-views tensor: (B, n_all, ...)[bi, ids]
        return LookupRecords(self.t[idx])

    @property
    def shape(self):
        return self.t.shape


def conv_top(x, wpack, bias, wpack_top, bias_top, relu=True, quad=False):
    """conv1x1(act(conv3x3(x (B,32,H,W)))) -> (B,H,W,32) channel-last, or (B,8,H,W,4) quad-planar with `quad`:
    FeatureNet's conv2.1 + toplayer as one launch (the 1x1 layer is a second stage of the 3x3 layer's workgroups).  Both
    packs from `pack_conv`."""
    B, C, H, W = x.shape
    assert C == 32
    out = torch.empty((B, 8, H, W, 4) if quad else (B, H, W, 32), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    with ktimer.region(f"conv_top[32->32->32,{H}x{W}]"):
        rc = lib.bmv_conv_top_fwd(dptr(x.contiguous(), "x"), dptr(wpack, "wpack"), dptr(bias, "bias"),
                                  dptr(wpack_top, "wpack_top"), dptr(bias_top, "bias_top"), dptr(out), B, H, W,
                                  _slope(relu, None), 3 if quad else 1, stream())
    _lib.check(rc, "conv_top_fwd")
    return out


def conv0_fused(x, w0, b0, wpack, bias, Cout, out=None):
    """relu(conv3x3(relu(conv3x3(x (B,3,H,W); w0 (8,3,3,3)) + b0); wpack) + bias): FeatureNet's first block (two
    ConvBnReLU, batch norm folded) as one launch; `wpack` / `bias` = pack_conv of the second layer."""
    B, _, H, W = x.shape
    assert x.shape[1] == 3 and tuple(w0.shape) == (8, 3, 3, 3)
    if out is None:
        out = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    x = x.contiguous()
    if ops.defer_table is not None:
        ops.defer_input(x)
    with ktimer.region(f"conv0_fused[3->8->{Cout},{H}x{W}]"):
        rc = lib.bmv_conv0_fused_fwd(dptr(x, "x"), dptr(w0, "w0"), dptr(b0, "b0"), dptr(wpack, "wpack"),
                                     dptr(bias, "bias"), dptr(out), B, Cout, H, W, 0.0, 0.0, stream())
    _lib.check(rc, "conv0_fused_fwd")
    return out


class VolumeRecords:
    """(B, D, h, w, 8) feature volume as the fused renderer reads it: a voxel's 8 channels in one 32-byte record,
    [ch 0 2 4 6 | ch 1 3 5 7] (include/bmv.h, bmv_conv_heads_fwd / bmv_render_args.vol_packed)."""
    ORDER = (0, 2, 4, 6, 1, 3, 5, 7, 8)      # output-channel order the 9-channel head conv's weights are packed in

    def __init__(self, t):
        assert t.dim() == 5 and t.shape[-1] == 8
        self.t = t

    @property
    def shape(self):
        return self.t.shape


def conv_heads_records(x, wpack, bias):
    """x (B,Cin,D,H,W) -> (VolumeRecords (B,D,H,W,8), depth logits (B,D,H,W)): feat_conv + depth_conv as one 3x3x3
    convolution whose epilogue writes the renderer's volume records (wpack / bias packed in VolumeRecords.ORDER)."""
    B, Cin, D, H, W = x.shape
    rec = torch.empty(B, D, H, W, 8, device=x.device, dtype=torch.float32)
    dp = torch.empty(B, D, H, W, device=x.device, dtype=torch.float32)
    x = x if x.is_contiguous() else x.contiguous()
    lib = _lib.load()
    with ktimer.region(f"conv[{Cin}->9 records,k3x3x3,s1,{D}x{H}x{W}]"):
        rc = lib.bmv_conv_heads_fwd(dptr(x, "conv input"), dptr(wpack, "wpack"), dptr(bias, "bias"), dptr(rec), dptr(dp),
                                    B, Cin, D, H, W, stream())
    _lib.check(rc, "conv_heads_fwd")
    return VolumeRecords(rec), dp


def fpn_smooth(fine, coarse, lat_weight, lat_bias, wpack, bias, Cout, out=None, rgb=None):
    """conv3x3(bilinear_x2(coarse, align_corners=True) + conv1x1(fine, lat_weight, lat_bias); wpack, bias) -> (B,Cout,H,W):
    FeatureNet's last top-down step and the smoothing conv that consumes it as ONE launch (the 32-channel
    full-resolution map between them is never written).  wpack / bias: `pack_conv` of the 3x3 layer (Cout <= 8)."""
    B, Cf, H, W = fine.shape
    C = coarse.shape[1]
    assert coarse.shape == (B, C, H // 2, W // 2)
    w = lat_weight.detach().reshape(C, Cf).contiguous()
    lib = _lib.load()
    packed = None
    if rgb is not None:
        # `rgb` (B,3,H,W): write the renderer's lookup records instead of the planar map; wpack / bias must have been
        # packed with the output channels in LookupRecords.EVEN_ODD order
        assert tuple(rgb.shape) == (B, 3, H, W) and Cout == 8
        packed = torch.empty(B, H, W, 12, device=fine.device, dtype=torch.float32)
    elif out is None:
        out = torch.empty(B, Cout, H, W, device=fine.device, dtype=torch.float32)
    if packed is not None:
        rgb = rgb.contiguous()
        if ops.defer_table is not None:
            ops.defer_input(rgb)
    with ktimer.region(f"fpn_smooth[{Cf}+{C}->{Cout},{H}x{W}]"):
        rc = lib.bmv_fpn_smooth_fwd(dptr(fine.contiguous(), "fine"), dptr(coarse.contiguous(), "coarse"), dptr(w, "w_lat"),
                                    dptr(lat_bias.detach().contiguous(), "b_lat"), dptr(wpack, "wpack"), dptr(bias, "bias"),
                                    None if packed is not None else dptr(out),
                                    dptr(rgb, "rgb") if packed is not None else None,
                                    dptr(packed) if packed is not None else None, B, Cf, C, Cout, H, W, 1.0, stream())
    _lib.check(rc, "fpn_smooth_fwd")
    return LookupRecords(packed) if packed is not None else out


def _xpair_octets(w):
    """weight (8, 8 n, 3, 3) float32 -> int32 [octet n][filter row 3][piece 3][lane 64][4]: the x-paired A operands of
    csrc/fpn_s.hip (lane = 16 kk + m: matrix row m = (output column parity r = m // 8, channel m % 8), input column slot kk
    of a column pair: weight[.., ky, kk - r] where 0 <= kk - r <= 2, else zero), every value split into three bf16 pieces."""
    nch = w.shape[1]
    w16 = torch.zeros(2, 8, nch, 3, 4, device=w.device, dtype=torch.float32)     # (r, c, ch, ky, slot kk): kx = kk - r
    w16[0, :, :, :, 0:3] = w
    w16[1, :, :, :, 1:4] = w
    # (m, o, i, ky, kk) -> (o, ky, kk, m, i): lane = 16 kk + m, the lane's 8 values = the octet's channels
    t = w16.reshape(16, nch // 8, 8, 3, 4).permute(1, 3, 4, 0, 2).reshape(nch // 8, 3, 64, 8).contiguous()
    return torch.stack(_split3_words(t), 2).contiguous()                          # (n, 3, 3, 64, 4)


def pack_conv0_s(w0, b0, w1, b1):
    """(w0b0, wsplit, bias) of bmv_conv0_s_fwd (csrc/fpn_s.hip): FeatureNet's first block, batch norm folded -- w0 (8,3,3,3),
    b0 (8): the first layer as [channel][27 weights | bias]; w1 (8,8,3,3), b1 (8): the second layer as one x-paired octet."""
    assert tuple(w0.shape) == (8, 3, 3, 3) and tuple(w1.shape) == (8, 8, 3, 3)
    w0b0 = torch.cat([w0.detach().float().reshape(8, 27), b0.detach().float().reshape(8, 1)], 1).contiguous()
    return w0b0, _xpair_octets(w1.detach().float())[0].contiguous(), b1.detach().float().contiguous()


def conv0_s(x, w0b0, wsplit, bias, out=None):
    """relu(conv3x3(relu(conv3x3(x (B,3,H,W)) + b0)) + bias) -> (B,8,H,W): FeatureNet's first block with the second layer on
    the bf16 matrix cores with three-piece fp32 operands (csrc/fpn_s.hip conv0_s_kernel; `pack_conv0_s`)."""
    B, _, H, W = x.shape
    assert x.shape[1] == 3
    if out is None:
        out = torch.empty(B, 8, H, W, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    assert wsplit.dtype == torch.int32 and wsplit.numel() == lib.bmv_conv0_s_wsplit_ints() and w0b0.numel() == 224
    x = x.contiguous()
    if ops.defer_table is not None:
        ops.defer_input(x)
    with ktimer.region(f"conv0_s[3->8->8,{H}x{W}]"):
        rc = lib.bmv_conv0_s_fwd(dptr(x, "x"), dptr(w0b0, "w0b0"), dptr(wsplit, "wsplit", torch.int32), dptr(bias, "bias"),
                                 dptr(out), B, H, W, 0.0, stream())
    _lib.check(rc, "conv0_s_fwd")
    return out


def pack_conv2d_s(weight, bias):
    """(wsplit, bias) of bmv_conv2d_s_fwd (csrc/conv2d_s.hip; include/bmv.h has the layout): weight (Cout % 16 == 0, Cin % 8 == 0,
    ks, ks) -> int32 [M tile][filter row][step][piece 3][lane 64][4]; lane = 16 kg + m holds the 8 channels of input octet o at
    filter column kx for output channel 16 tile + m, (o, kx) = divmod(4 step + kg, ks), every value as three bf16 pieces."""
    Cout, Cin, ks, ks2 = weight.shape
    lib = _lib.load()
    assert ks == ks2 and lib.bmv_conv2d_s_wsplit_ints(Cin, Cout, ks, 2 if ks == 5 else 1) > 0, "shape not covered by conv2d_s"
    noct, nmt = Cin // 8, Cout // 16
    npair = noct * ks
    nstep = (npair + 3) // 4
    w = weight.detach().float().reshape(nmt, 16, noct, 8, ks, ks)                     # (tile, m, octet, c, ky, kx)
    w = w.permute(0, 1, 4, 2, 5, 3).reshape(nmt, 16, ks, npair, 8)                    # (tile, m, ky, pair, c)
    wz = torch.zeros(nmt, 16, ks, nstep * 4, 8, device=w.device, dtype=torch.float32)
    wz[:, :, :, :npair] = w
    t = wz.reshape(nmt, 16, ks, nstep, 4, 8).permute(0, 2, 3, 4, 1, 5).reshape(nmt, ks, nstep, 64, 8).contiguous()
    wsplit = torch.stack(_split3_words(t), 3).contiguous()                            # (tile, ky, step, 3, 64, 4)
    assert wsplit.numel() == lib.bmv_conv2d_s_wsplit_ints(Cin, Cout, ks, 2 if ks == 5 else 1)
    b = bias.detach().float().contiguous() if bias is not None else torch.zeros(Cout, device=w.device, dtype=torch.float32)
    return wsplit, b


class SplitRecords:
    """A feature map as SPLIT RECORDS (csrc/conv2d_s.hip, include/bmv.h): `data` (B, C/8, 3, H, W, 4) int32 -- the 8 channels
    of an octet at a pixel as 8 bf16 (16 bytes) per piece, hi + mid + lo = the fp32 value exactly.  Written by the producing
    layer's epilogue and staged by the consuming layer with LDS-DMA.  `shape` is the logical planar shape, `to_planar()` the
    fp32 tensor (exact)."""
    __slots__ = ("data",)

    def __init__(self, data):
        assert data.dtype == torch.int32 and data.dim() == 6 and data.shape[2] == 3 and data.shape[-1] == 4 and data.is_contiguous()
        self.data = data

    @staticmethod
    def empty(B, C, H, W, device):
        assert C % 8 == 0
        return SplitRecords(torch.empty(B, C // 8, 3, H, W, 4, device=device, dtype=torch.int32))

    @staticmethod
    def from_planar(x):
        """(B, C % 8 == 0, H, W) float32 -> records (the arithmetic of the kernels' epilogue; tests and first layers)."""
        B, C, H, W = x.shape
        t = x.float().reshape(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()        # (B, octet, H, W, 8)
        return SplitRecords(torch.stack(_split3_words(t), 2).contiguous())

    @property
    def shape(self):
        B, O, _, H, W, _ = self.data.shape
        return torch.Size((B, O * 8, H, W))

    @property
    def device(self):
        return self.data.device

    def to_planar(self):
        B, O, _, H, W, _ = self.data.shape
        w = self.data.view(torch.int16).reshape(B, O, 3, H, W, 8).to(torch.int32) << 16      # bf16 -> the fp32 bit pattern
        f = w.view(torch.float32)
        return ((f[:, :, 2] + f[:, :, 1]) + f[:, :, 0]).permute(0, 1, 4, 2, 3).reshape(B, O * 8, H, W)


def conv2d_s(x, wsplit, bias, Cout, ks, stride, relu=False, slope=None, out=None, records=False):
    """act(conv2d(x (B,Cin,H,W), k = ks, stride, padding ks // 2) + bias) -> (B,Cout,H/stride,W/stride) on the bf16 matrix cores
    with three-piece fp32 operands (csrc/conv2d_s.hip; `pack_conv2d_s`): FeatureNet's 5x5 stride-2 and 3x3 encoder layers.
    `x` may be a SplitRecords (staged by LDS-DMA: no split in this layer); records=True: the result as SplitRecords,
    records="both": (planar, SplitRecords)."""
    rin = isinstance(x, SplitRecords)
    B, Cin, H, W = x.shape
    Ho, Wo = H // stride, W // stride
    dev = x.device
    want_planar = records is False or records == "both"
    if want_planar and out is None:
        out = torch.empty(B, Cout, Ho, Wo, device=dev, dtype=torch.float32)
    assert out is None or (out.shape == (B, Cout, Ho, Wo) and out.is_contiguous())
    rec = SplitRecords.empty(B, Cout, Ho, Wo, dev) if records else None
    lib = _lib.load()
    assert wsplit.dtype == torch.int32 and wsplit.numel() == lib.bmv_conv2d_s_wsplit_ints(Cin, Cout, ks, stride)
    if not rin:
        x = x if x.is_contiguous() else x.contiguous()
    with ktimer.region(f"conv2d_s[{Cin}->{Cout},k{ks}s{stride},{H}x{W}{',rec' if rin else ''}]"):
        rc = lib.bmv_conv2d_s_fwd(None if rin else dptr(x, "x"), dptr(x.data, "x records", torch.int32) if rin else None,
                                  dptr(wsplit, "wsplit", torch.int32), dptr(bias, "bias"), dptr(out) if want_planar else None,
                                  dptr(rec.data, "out records", torch.int32) if rec is not None else None, B, Cin, H, W, Cout, ks, stride,
                                  _slope(relu, slope), stream())
    _lib.check(rc, "conv2d_s_fwd")
    if records == "both":
        return out, rec
    return rec if records else out


def pack_fpn_smooth_s(smooth_weight, smooth_bias, lat_weight, lat_bias, order=None):
    """(wsplit, btab) of bmv_fpn_smooth_s_fwd (csrc/fpn_s.hip, include/bmv.h): smooth0 (8, 32, 3, 3) + bias (8) or None,
    lat0 (32, 8, 1, 1) + bias (32).  The lateral 1x1 convolution is folded into the 3x3 weights in float64 (smooth0 is
    linear): octets 0..3 = smooth0 on bilinear_x2(p1), octet 4 = smooth0 . lat0 on c0, btab[row case][column case] =
    smooth0's bias + smooth0 applied to the lateral bias through the taps INSIDE the image (cases: first / interior / last
    row resp. column).  `order`: output-channel permutation (LookupRecords.EVEN_ODD for the renderer's records)."""
    Ws = smooth_weight.detach().double()
    bs = smooth_bias.detach().double() if smooth_bias is not None else torch.zeros(Ws.shape[0], dtype=torch.float64, device=Ws.device)
    assert tuple(Ws.shape) == (8, 32, 3, 3) and tuple(lat_weight.shape[:2]) == (32, 8)
    if order is not None:
        Ws, bs = Ws[list(order)], bs[list(order)]
    Wl = lat_weight.detach().double().reshape(32, 8)
    bl = lat_bias.detach().double()
    # (broadcast products + sums, not einsum / matmul: the pack runs on the parameters' device, and the GPU suite asserts
    # that no GEMM library kernel is launched on the path)
    Wc = (Ws[:, :, None] * Wl[None, :, :, None, None]).sum(1)        # (8, 8, 3, 3): smooth0 . lat0
    T = (Ws * bl[None, :, None, None]).sum(1)                        # (8, 3, 3): what a tap adds through the lateral bias
    Wfull = torch.cat([Ws, Wc], 1).float()                           # (8, 40, 3, 3)
    wsplit = _xpair_octets(Wfull)                                     # (5, 3, 3, 64, 4)
    valid = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}                      # taps inside the image by case (first, interior, last)
    btab = torch.stack([torch.stack([bs + T[:, list(valid[yc])][:, :, list(valid[xc])].sum((1, 2)) for xc in range(3)])
                        for yc in range(3)]).float().contiguous()     # (3, 3, 8)
    return wsplit, btab


def fpn_smooth_s(fine, coarse, wsplit, btab, rgb=None):
    """smooth0(bilinear_x2(coarse, align_corners=True) + lat0(fine)) on the bf16 matrix cores with three-piece fp32
    operands and the lateral convolution folded into the weights (csrc/fpn_s.hip; `pack_fpn_smooth_s`): fine (B,8,H,W),
    coarse (B,32,H/2,W/2) -> (B,8,H,W), or with `rgb` (B,3,H,W) the renderer's LookupRecords (weights packed with
    order=LookupRecords.EVEN_ODD)."""
    B, Cf, H, W = fine.shape
    assert Cf == 8 and tuple(coarse.shape) == (B, 32, H // 2, W // 2) and H % 2 == 0 and W % 2 == 0
    lib = _lib.load()
    assert wsplit.dtype == torch.int32 and wsplit.numel() == lib.bmv_fpn_smooth_s_wsplit_ints() and btab.numel() == 72
    packed = out = None
    if rgb is not None:
        assert tuple(rgb.shape) == (B, 3, H, W)
        packed = torch.empty(B, H, W, 12, device=fine.device, dtype=torch.float32)
        rgb = rgb.contiguous()
        if ops.defer_table is not None:
            ops.defer_input(rgb)
    else:
        out = torch.empty(B, 8, H, W, device=fine.device, dtype=torch.float32)
    with ktimer.region(f"fpn_smooth_s[8+32->8,{H}x{W}]"):
        rc = lib.bmv_fpn_smooth_s_fwd(dptr(fine.contiguous(), "fine"), dptr(coarse.contiguous(), "coarse"),
                                      dptr(wsplit, "wsplit", torch.int32), dptr(btab, "btab"),
                                      dptr(out) if out is not None else None, dptr(rgb, "rgb") if packed is not None else None,
                                      dptr(packed) if packed is not None else None, B, H, W, 1.0, stream())
    _lib.check(rc, "fpn_smooth_s_fwd")
    return LookupRecords(packed) if packed is not None else out


def conv_fwd(x, wpack, bias, Cout, kd, k, stride=1, relu=False, skip=None, channels_last=False, out=None, slope=None):
    """x (B,Cin,H,W) or (B,Cin,D,H,W) planar -> act(conv(x) + bias) + skip, planar or channel-last
    ((B,Ho,Wo,Cout) / (B,Do,Ho,Wo,Cout)), or -- channels_last="quad" -- quad-planar (B,Cout/4,[Do,]Ho,Wo,4)."""
    qin = isinstance(x, ops.QuadVolume)
    if qin and not (kd == 3 and k == 3 and stride == 2 and x.shape[1] < 16 and Cout <= 16):
        x, qin = x.to_planar().contiguous(), False       # (only the regularisers' stride-2 layer stages quad records)
    is3d = qin or x.dim() == 5
    B, Cin = x.shape[:2]
    D = x.shape[2] if is3d else 1
    H, W = x.shape[-2:]
    if qin:
        x = x.data
    p, pd = k // 2, kd // 2
    Do, Ho, Wo = (D + 2 * pd - kd) // stride + 1, (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1
    quad = channels_last == "quad"          # (B, Cout/4, [Do,] Ho, Wo, 4): the plane sweep's source layout
    if quad:
        shape = (B, Cout // 4, Do, Ho, Wo, 4) if is3d else (B, Cout // 4, Ho, Wo, 4)
    elif channels_last:
        shape = (B, Do, Ho, Wo, Cout) if is3d else (B, Ho, Wo, Cout)
    else:
        shape = (B, Cout, Do, Ho, Wo) if is3d else (B, Cout, Ho, Wo)
    if out is None:
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    assert out.shape == shape and out.is_contiguous()
    if skip is not None:
        assert skip.shape == shape and skip.is_contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    lib = _lib.load()
    with ktimer.region(f"conv[{Cin}->{Cout},k{kd}x{k}x{k},s{stride},{D}x{H}x{W}]"):
        rc = lib.bmv_conv_fwd(dptr(x, "conv input"), dptr(wpack, "wpack"), dptr(bias, "bias"),
                              dptr(skip) if skip is not None else None, dptr(out), B, Cin, D, H, W, Cout, kd, k, stride,
                              _slope(relu, slope), (3 if quad else int(bool(channels_last))) | (16 if qin else 0), stream())
    _lib.check(rc, "conv_fwd")
    return out
