import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from boostmvsnerfs_amd import _lib, convnet, ops
for (B, Cin, Cout, pair, sp) in [(1, 16, 8, True, (8, 256, 320)), (1, 16, 8, True, (8, 64, 64)), (1, 8, 8, True, (8, 256, 320)), (1, 16, 8, True, (4, 256, 320))]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, *sp, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    D_, H_, W_ = sp
    qv = ops.QuadVolume(x.view(B, Cin // 4, 4, D_, H_, W_).permute(0, 1, 3, 4, 5, 2).contiguous())
    ws, bs, pr = convnet.pack_conv_c4s(w, b, pair)
    want = F.conv3d(x, w, b, 1, 1)
    for rw, tz in ((4, 4), (2, 4), (4, 2)):
        _lib.set_tuning("BMV_CONV_C4S_RW", rw), _lib.set_tuning("BMV_CONV_C4S_TZ", tz)
        for rep in range(3):
            got = convnet.conv_c4s_fwd(qv, ws, bs, pr, Cout)
            d = (got - want).abs()
            bad = d > 1e-3
            print(f"{Cin}->{Cout} {sp} RW{rw} TZ{tz} rep{rep}: max err {float(d.max()):.3e}; bad voxels {int(bad.sum())}; per-plane bad {[int(bad[:, :, z].sum()) for z in range(D_)]}; per-channel bad {[int(bad[:, c].sum()) for c in range(Cout)]}")
            if bad.any():
                idx = bad.nonzero()
                print("   first bad", idx[:3].tolist(), "y range", int(idx[:, 3].min()), int(idx[:, 3].max()), "x range", int(idx[:, 4].min()), int(idx[:, 4].max()),
                      "ytile hist", torch.bincount(idx[:, 3] // 16)[:20].tolist())
