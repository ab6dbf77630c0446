#!/usr/bin/env python
"""Same-box A/B of the training step under environment switches: every configuration runs `bench.py --mode train` in its own
process, the list is walked `--rounds` times (alternating, so that a drifting box shows up as a spread, not as a winner).
  usage: tools/train_ab.py [--rounds 2] [--steps 20] [--keys bwd.bnact_bwd,fwd.bn_stats] base: u2off:Y6_BNBWD_U2=0 ...
Prints one line per run: name, ms per step, the chosen breakdown entries, the bits of the last loss."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--keys", default="")
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    ap.add_argument("configs", nargs="+", help="name:K=V,K=V")
    a = ap.parse_args()
    keys = [k for k in a.keys.split(",") if k]
    for r in range(a.rounds):
        for c in a.configs:
            name, _, envs = c.partition(":")
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", str(a.steps), "--warmup", str(a.warmup),
                   "--no-autotune"] + a.extra.split()
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                print(f"{name} FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)
                continue
            d = json.loads(line[-1])
            br = d.get("breakdown", {})
            sel = " ".join(f"{k}={br[k]['ms']:.3f}" for k in keys if k in br)
            print(f"{name:12s} {d['ms_per_step']:.3f} ms  {sel}  bits {d['loss']['bits'][-1]}  attempts {d.get('supervisor', {}).get('attempts')}",
                  flush=True)


if __name__ == "__main__":
    main()
