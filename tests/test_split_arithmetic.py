"""The arithmetic of the renderer's bf16 x 3 chains (csrc/mlp.hpp, BMV_SPLIT_CHAIN2), restated in numpy on the CPU: an fp32
value is cut by ROUND TO NEAREST EVEN (v_cvt_pk_bf16_f32) into hi = rn(x), mid = rn(x - hi), lo = x - hi - mid; a product
keeps hi hi + hi mid + mid hi + hi lo + lo hi + mid mid and drops mid lo + lo mid + lo lo.  Checked here: the three
pieces are bf16 numbers and reproduce the fp32 value EXACTLY, every kept partial product is exact in fp32, and what is
dropped is at most 2^-23 of the product -- one fp32 rounding of it."""
import numpy as np


def _rn_bf16(x):
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _pieces(x):
    hi = _rn_bf16(x)
    r1 = x - hi
    mid = _rn_bf16(r1)
    lo = r1 - mid
    return hi, mid, lo


def _values(seed, n):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)


def test_three_bf16_pieces_are_the_fp32_value():
    x = np.concatenate([_values(0, 1 << 16), np.array([0.0, -0.0, 1.0, -1.0, np.pi, 1.0 + 2.0 ** -23, 255.0 / 256.0, 1e30, -1e-30], np.float32)])
    hi, mid, lo = _pieces(x)
    for p in (hi, mid, lo):                      # every piece is a bf16 number (low 16 bits clear): lo needs no rounding
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert float((np.abs(mid[nz]) / np.abs(x[nz])).max()) <= 2.0 ** -8 and float((np.abs(lo[nz]) / np.abs(x[nz])).max()) <= 2.0 ** -16


def test_kept_products_are_exact_and_the_dropped_ones_are_one_fp32_rounding():
    a, b = _values(1, 1 << 16), _values(2, 1 << 16)
    ah, am, al = _pieces(a)
    bh, bm, bl = _pieces(b)
    kept = [(ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)]
    for u, v in kept:                            # 8-bit x 8-bit significands: 16 bits, exact in fp32 (the MFMA's products)
        assert np.array_equal((u * v).astype(np.float64), u.astype(np.float64) * v.astype(np.float64))
    full = a.astype(np.float64) * b.astype(np.float64)
    got = sum(u.astype(np.float64) * v.astype(np.float64) for u, v in kept)
    rel = np.abs(full - got) / np.abs(full)
    assert float(rel.max()) <= 2.0 ** -23 * 1.01, float(rel.max())
    assert float(np.median(rel)) < 2.0 ** -26


def test_split_records_round_trip_is_exact():
    """convnet.SplitRecords (the activation format between two bf16 x 3 encoder layers, csrc/conv2d_s.hip): hi / mid by
    TRUNCATION, lo = the rest rounded to bf16 (at most 8 bits are left: exact) -- from_planar -> to_planar returns the fp32
    tensor bit for bit, and the records are laid out (B, C/8, piece, H, W, 8 bf16)."""
    import torch
    from boostmvsnerfs_amd.convnet import SplitRecords
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 6, 10, generator=g) * 10.0 ** torch.randint(-6, 6, (2, 16, 6, 10), generator=g).float()
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0 + 2.0 ** -23, -255.0 / 256.0])
    r = SplitRecords.from_planar(x)
    assert tuple(r.data.shape) == (2, 2, 3, 6, 10, 4) and r.shape == x.shape
    assert torch.equal(r.to_planar(), x)
    # channel 8 o + 2 i (+ 1) of pixel (y, x) sits in the low (high) half of word i of record (o, piece, y, x)
    hi = r.data[1, 1, 0, 3, 7, 2]
    assert (int(hi) & 0xFFFF) == (int(x[1, 8 + 4, 3, 7].view(torch.int32)) >> 16) & 0xFFFF
    assert ((int(hi) >> 16) & 0xFFFF) == (int(x[1, 8 + 5, 3, 7].view(torch.int32)) >> 16) & 0xFFFF


def test_conv2d_s_weight_pack_is_the_documented_layout():
    """convnet.pack_conv2d_s (include/bmv.h, bmv_conv2d_s_fwd): int32 [M tile][filter row][step][piece 3][lane 64][4]; lane = 16 kg +
    m holds the 8 channels of input octet o at filter column kx for output channel 16 tile + m, (o, kx) = divmod(4 step + kg,
    ks); pairs past the end are zeros; hi + mid + lo of the three pieces is the fp32 weight exactly.  (Host logic: runs
    without a GPU; the library is loaded for the word count only.)"""
    import torch
    from boostmvsnerfs_amd import convnet
    g = torch.Generator().manual_seed(1)
    for cout, cin, ks in ((16, 8, 5), (32, 16, 5), (16, 16, 3), (32, 32, 3)):
        w = torch.randn(cout, cin, ks, ks, generator=g) * 10.0 ** torch.randint(-3, 3, (cout, cin, ks, ks), generator=g).float()
        ws, b = convnet.pack_conv2d_s(w, None)
        npair, nstep = (cin // 8) * ks, ((cin // 8) * ks + 3) // 4
        assert tuple(ws.shape) == (cout // 16, ks, nstep, 3, 64, 4) and ws.dtype == torch.int32 and float(b.abs().max()) == 0.0
        halves = ws.view(torch.int16).reshape(cout // 16, ks, nstep, 3, 64, 8).to(torch.int32) << 16      # bf16 -> fp32 bits
        val = halves.view(torch.float32)
        total = (val[:, :, :, 2] + val[:, :, :, 1]) + val[:, :, :, 0]                                      # (tile, ky, step, lane, c)
        for tile in range(cout // 16):
            for step in range(nstep):
                for kg in range(4):
                    pi = 4 * step + kg
                    got = total[tile, :, step, 16 * kg:16 * kg + 16]                                        # (ky, m, c)
                    if pi >= npair:
                        assert float(got.abs().max()) == 0.0
                        continue
                    o, kx = divmod(pi, ks)
                    want = w[16 * tile:16 * tile + 16, 8 * o:8 * o + 8, :, kx].permute(2, 0, 1)            # (ky, m, c)
                    assert torch.equal(got, want), (cout, cin, ks, tile, step, kg)
