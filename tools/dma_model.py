"""Where the time of a 3x3 stride-1 LDS-DMA launch goes, from the launch geometry alone (no GPU needed).

Mirrors the host-side sizing of csrc/conv_mfma.hip (choose_tile, build_launch) and csrc/conv_dma.hip (launch_dma: persistent
grid = CUs x resident blocks) for the 16-channel-chunk forms dma_c2p2 / dma_c2p1, and prices every launch of a measured op
table (`bench.py --ops-json`, e.g. profiles/r02/bench_infer_ops_r02d.json) with

    t_model = rounds x chunks x chunk_cycles / clock          (matrix pipe shared by the blocks resident on a CU)

where chunk_cycles = resident_waves_per_SIMD x 9 taps x CF x PF MFMAs x 32 cycles / e_chunk, e_chunk = the in-chunk matrix-pipe
efficiency of the s_memtime traces (nine taps 2 700 cycles for 2 304 of MFMA, + 460 at the barrier -> 0.73).
and with the fill floor t_fill = items x chunks x (halo + weight image bytes) / (18.7 B/clk/CU x 256 CUs), the LDS-DMA rate
measured with the MFMAs compiled out.  Printed per layer: tile padding, round quantisation (items / (grid x rounds)), both
floors, the weight share of the fill bytes, the measured time and its ratio to max(floors) (1.0 = perfectly overlapped) and to
their sum (1.0 = not overlapped at all).

    python tools/dma_model.py profiles/r02/bench_infer_ops_r02d.json
    python tools/dma_model.py --what-if        # both floors for every variant, incl. the ones not measured yet
"""
import json
import math
import sys

CUS, CLOCK_GHZ, LDS_KB = 256, 2.4, 160
E_CHUNK = 2304.0 / (2700.0 + 460.0)
FILL_B_PER_CLK_CU = 18.7      # measured: requests + barriers only, 371 MB in 38.5 us


def cdiv(a, b):
    return -(-a // b)


def halo_cap(bp):
    return 208 if bp <= 128 else 352 if bp <= 256 else 672 if bp <= 512 else 1216


def choose_tile(Ho, Wo, bp, cap, tw_mult=16):
    best, bt = -1.0, (1, 1)
    for TW in range(1, min(Wo, bp) + 1):
        TH = min(bp // TW, Ho)
        while TH >= 1 and (TH + 2) * (TW + 2) > cap:
            TH -= 1
        if TH < 1:
            continue
        tiles = cdiv(Ho, TH) * cdiv(Wo, TW)
        eff = Ho * Wo / (tiles * bp)
        halo = TH * TW / ((TH + 2) * (TW + 2))
        score = eff * (0.85 + 0.15 * halo) * (1.0 if TW % tw_mult == 0 else 0.95)
        if score > best + 1e-9:
            best, bt = score, (TH, TW)
    return bt


VARIANTS = {"dma_c2p2": dict(cf=2, pf=2, nw=4), "dma_c2p1": dict(cf=2, pf=1, nw=4), "dma8_c2p2": dict(cf=2, pf=2, nw=8),
            # written after round 2's last GPU visit (not measured yet): --what-if prices them with the same two rates
            "dma_c2p4": dict(cf=2, pf=4, nw=4), "dma8_c4p1": dict(cf=4, pf=1, nw=8), "dmar8_c2p2": dict(cf=2, pf=2, nw=8, wres=1)}


def launch_geometry(B, C, K, H, W, variant):
    v = VARIANTS[variant]
    bp = v["nw"] * v["pf"] * 32
    TH, TW = choose_tile(H, W, bp, halo_cap(bp))
    tiles = B * cdiv(H, TH) * cdiv(W, TW)
    ncb = cdiv(cdiv(K, 32), v["cf"])
    nids = tiles if ncb == 1 else cdiv(tiles, 8) * 8 * ncb
    nhp = cdiv(2 * (TH + 2) * (TW + 2), 64)
    wres = v.get("wres", 0)
    if wres and C > 64:
        return None
    if v["cf"] > cdiv(K, 32):
        return None
    lds = (2 * nhp + cdiv(C, 16) * 9 * v["cf"]) * 1024 + 8 * v["cf"] * 32 * 4 if wres else 2 * (nhp + 9 * v["cf"]) * 1024 + 8 * v["cf"] * 32 * 4
    bpc = max(1, min(LDS_KB * 1024 // lds, 32 // v["nw"]))
    grid = min(nids, CUS * bpc - (CUS * bpc) % 8)
    rounds = cdiv(nids, grid)
    chunks = cdiv(C, 16)
    waves_per_simd = bpc * v["nw"] / 4.0 if grid >= CUS * bpc - 8 else max(1.0, nids / CUS) * v["nw"] / 4.0
    chunk_cycles = waves_per_simd * 9 * v["cf"] * v["pf"] * 32 / E_CHUNK
    fill_bytes = nids * chunks * (nhp + (0 if wres else 9 * v["cf"])) * 1024 + (grid * chunks * 9 * v["cf"] * 1024 if wres else 0)
    t_fill_us = fill_bytes / (FILL_B_PER_CLK_CU * CUS * CLOCK_GHZ * 1e3)
    return dict(tile=(TH, TW), tiles=tiles, items=nids, lds=lds, bpc=bpc, grid=grid, rounds=rounds, t_fill_us=t_fill_us,
                w_share=9 * v["cf"] / (nhp + 9 * v["cf"]),
                e_pad=B * H * W / (tiles * bp), e_round=nids / (grid * rounds),
                t_model_us=rounds * chunks * chunk_cycles / (CLOCK_GHZ * 1e3))


def main(path):
    rows = json.load(open(path))["rows"]
    print(f"{'op':>3} {'layer':>18} {'variant':>9} {'tile':>7} {'items':>6} {'grid':>5} {'rnd':>3} {'e_pad':>5} {'e_rnd':>5} "
          f"{'ideal':>6} {'mfma':>6} {'fill':>6} {'wgt%':>4} {'meas':>6} {'m/max':>5} {'m/sum':>5}  (us)")
    tot = dict(ideal=0.0, model=0.0, meas=0.0, fill=0.0)
    for r in rows:
        if r["kind"] != "conv" or r["ksize"] != 3 or r["stride"] != 1 or r["variant"] not in VARIANTS:
            continue
        # flops = 2 * B*H*W * 9*C*K ; bytes (fp16) = 2 * (B*H*W*(C+K) + 9*C*K): solve with C == K first, then the 2:1 head convs
        flops, byts = r["flops"], r["bytes"]
        found = None
        for H in (160, 80, 40, 20):
            px = 32 * H * H
            for C in (32, 64, 128, 256, 512):
                for K in (C, 2 * C, C // 2):
                    if abs(18.0 * px * C * K - flops) / flops < 1e-6 and abs(2 * (px * (C + K) + 9 * C * K) - byts) / byts < 0.05:
                        found = (C, K, H)
        if not found:
            print(f"{r['op']:>3} (shape not recognised: {flops:.3e} flop, {byts:.3e} B)")
            continue
        C, K, H = found
        g = launch_geometry(32, C, K, H, H, r["variant"])
        ideal = flops / 2.5e15 * 1e6
        meas = r["ms"] * 1e3
        tot["ideal"] += ideal
        tot["model"] += g["t_model_us"]
        tot["meas"] += meas
        tot["fill"] += g["t_fill_us"]
        print(f"{r['op']:>3} {f'{C}->{K}@{H}x{H}':>18} {r['variant'][4:]:>9} {g['tile'][0]:>3}x{g['tile'][1]:<3} {g['items']:>6} "
              f"{g['grid']:>5} {g['rounds']:>3} {g['e_pad']:>5.2f} {g['e_round']:>5.2f} {ideal:>6.1f} {g['t_model_us']:>6.1f} "
              f"{g['t_fill_us']:>6.1f} {100 * g['w_share']:>4.0f} {meas:>6.1f} {meas / max(g['t_model_us'], g['t_fill_us']):>5.2f} "
              f"{meas / (g['t_model_us'] + g['t_fill_us']):>5.2f}")
    print(f"sum: ideal {tot['ideal']:.0f} us, matrix-pipe model {tot['model']:.0f} us, fill floor {tot['fill']:.0f} us, measured "
          f"{tot['meas']:.0f} us (in-chunk efficiency {E_CHUNK:.2f}, fill {FILL_B_PER_CLK_CU} B/clk/CU)")


def what_if():
    """max(matrix floor, fill floor) of every variant on the seven 3x3 stride-1 shapes of YOLOv6-S b32 (byte pricing of the fill)."""
    shapes = [(64, 64, 160), (128, 128, 80), (256, 256, 40), (512, 512, 20), (64, 64, 80), (128, 128, 40), (256, 256, 20),
              (64, 128, 80), (128, 256, 40), (256, 512, 20)]
    print(f"{'layer':>16} " + " ".join(f"{v[3:]:>12}" for v in VARIANTS))
    for C, K, H in shapes:
        cells = []
        for v in VARIANTS:
            g = launch_geometry(32, C, K, H, H, v)
            cells.append(f"{'-':>12}" if g is None else f"{g['t_model_us']:5.1f}|{g['t_fill_us']:5.1f}".rjust(12))
        print(f"{f'{C}->{K}@{H}':>16} " + " ".join(cells))
    print("(matrix floor | fill floor, us; the measured launches sit at 1.2-1.7 x the larger one)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--what-if":
        what_if()
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r02/bench_infer_ops_r02d.json")
