"""Debug of conv_pw.hip: a 1x1 conv whose output names the (pixel, channel) each element was computed from.
x[p][c] = p (pixel index) or c; w = selector.  Prints where the kernel's output departs from the expectation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gpu_utils as G
from yolov6_amd.engine import TRef

names = G.variant_names()
vname = sys.argv[1] if len(sys.argv) > 1 else "pw_c4p2"
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
P = int(sys.argv[4]) if len(sys.argv) > 4 else 64
v = names.index(vname)
for mode in ("pixel", "channel", "random"):
    if mode == "pixel":      # out[p][co] = p
        x = torch.arange(P).view(1, 1, P, 1).expand(1, 1, P, Cin).clone().float()
        w = torch.zeros(Cout, Cin, 1, 1); w[:, 0] = 1.0
    elif mode == "channel":  # out[p][co] = co % Cin
        x = torch.arange(Cin).view(1, 1, 1, Cin).expand(1, 1, P, Cin).clone().float()
        w = torch.zeros(Cout, Cin, 1, 1)
        for co in range(Cout): w[co, co % Cin] = 1.0
    else:
        g = torch.Generator().manual_seed(0)
        x = torch.rand((1, 1, P, Cin), generator=g) * 2 - 1
        w = torch.randn((Cout, Cin, 1, 1), generator=g) / Cin ** 0.5
    xr = TRef(x.half().to(G.DEV), 1, 1, P, Cin, Cin, 0)
    b = torch.zeros(Cout)
    o, _ = G.run_conv(xr, w, b, 1, None, v)
    got = o.to_nhwc_tensor().float().cpu().view(P, Cout)
    ref = (x.half().float().view(P, Cin) @ w.half().float().view(Cout, Cin).t())
    bad = (got - ref).abs() > 1e-2 * ref.abs().clamp(min=1)
    print(f"== {vname} Cin {Cin} Cout {Cout} P {P} mode {mode}: {int(bad.sum())} / {bad.numel()} wrong")
    if bad.any() and mode != "random":
        torch.set_printoptions(linewidth=250, edgeitems=100, threshold=100000)
        print("got[:, ::8] rows 0..min(P,40):"); print(got[:min(P, 40), ::8].to(torch.int32))
    elif bad.any():
        print("rows wrong:", bad.any(1).nonzero().flatten().tolist()[:80]); print("cols wrong:", bad.any(0).nonzero().flatten().tolist()[:80])
