"""Pins ds_read_b64_tr_b16's mapping on the GPU box: prints, for two address patterns, which LDS half every (lane, element) got.
Pattern A: lane l reads at byte 8*l (the guide's contiguous form).  Pattern B: the wgrad form - a [pixel][32 channel] image with
64-byte pixel pitch; 16-lane group g: lanes 0-15 / 16-31 channels 0-15 / 16-31 of pixels p0..p0+3, lanes 32-63 the same of
pixels p1..p1+3; lane i of a group passes &img[p + i/4][cb + 4*(i%4)]."""
import ctypes as C
import json
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "tr16_probe.so"))
lib.tr16_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")


def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device=dev)
    out = torch.zeros(256, dtype=torch.int16, device=dev)
    rc = lib.tr16_probe(a.data_ptr(), out.data_ptr(), 4096, None)
    torch.cuda.synchronize()
    assert rc == 0, rc
    return out.cpu().view(64, 4).tolist()


res = {}
A = run([8 * l for l in range(64)])
res["contiguous"] = A
exp_guide = [[(l & 15) + j * 16 + (l >> 4) * 64 for j in range(4)] for l in range(64)]
res["contiguous_matches_guide"] = A == exp_guide
p0, p1 = 8, 40
addr = []
for l in range(64):
    g, i = l >> 4, l & 15
    p = (p0 if g < 2 else p1) + i // 4
    cb = 16 * (g & 1)
    addr.append(p * 64 + (cb + 4 * (i % 4)) * 2)
B = run(addr)
# expectation: lane l gets channel c = 16*(g&1) + i of pixels p..p+3  -> half index = pixel*32 + c
exp = [[((p0 if (l >> 4) < 2 else p1) + j) * 32 + 16 * ((l >> 4) & 1) + (l & 15) for j in range(4)] for l in range(64)]
res["wgrad_form"] = B
res["wgrad_form_matches"] = B == exp
print(json.dumps({k: v for k, v in res.items() if k.endswith("matches") or k.endswith("guide")}))
if not (res["contiguous_matches_guide"] and res["wgrad_form_matches"]):
    print("contiguous:", A[:20])
    print("wgrad form:", B[:20], "expected", exp[:20])
json.dump(res, open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout)
