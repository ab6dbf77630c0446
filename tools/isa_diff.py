#!/usr/bin/env python3
"""Device code of every kernel of yolov6_amd/csrc/*.hip at a git ref against the working tree, function by function
(labels and comments normalised).  No GPU needed: hipcc cross-compiles.  Used before committing host-side or gated changes made
without a GPU visit: the kernels the last visit tested must come out instruction for instruction.

    python tools/isa_diff.py [ref]          # default HEAD; exit code 1 if a kernel's code changed or disappeared
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolov6_amd.csrc import build as B  # noqa: E402


def device_asm(csrc, inc, name, flags, out):
    cmd = [B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-I" + inc, "-I" + csrc, "--cuda-device-only", "-S",
           os.path.join(csrc, name), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{r.stderr[-2000:]}")


def functions(path):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if cur:
                out[cur] = buf
            cur, buf = m.group(1), []
        elif cur is not None:
            if line.startswith("\t.amdhsa_kernel"):
                out[cur], cur, buf = buf, None, []
            else:
                t = re.sub(r"\.Lfunc_\w+", "", re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", line).strip()))
                t = re.sub(r"\.Lpost_getpc\d+", ".Lpost_getpc", t)      # (long-branch labels are numbered per file)
                if t:
                    buf.append(t)
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
    tmp = tempfile.mkdtemp(prefix="isa_diff_")
    wt = os.path.join(tmp, "wt")
    subprocess.run(["git", "-C", ROOT, "worktree", "add", "-q", "--detach", wt, ref], check=True)
    try:
        jobs = []
        for side, base in (("old", wt), ("new", ROOT)):
            os.makedirs(os.path.join(tmp, side))
            csrc = os.path.join(base, "yolov6_amd", "csrc")
            for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
                n = os.path.basename(f)
                jobs.append((csrc, os.path.join(base, "include"), n, B.SOURCES.get(n, []), os.path.join(tmp, side, n[:-4] + ".s")))
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
            list(ex.map(lambda j: device_asm(*j), jobs))
        bad = 0
        for f in sorted(glob.glob(os.path.join(tmp, "new", "*.s"))):
            n = os.path.basename(f)
            old_p = os.path.join(tmp, "old", n)
            a = functions(old_p) if os.path.exists(old_p) else {}
            b = functions(f)
            diff = [k for k in a if k in b and a[k] != b[k]]
            gone = [k for k in a if k not in b]
            new = [k for k in b if k not in a]
            print(f"{n[:-2]:12s} {len(b):3d} kernels   changed {len(diff)}   gone {len(gone)}   new {len(new)}")
            for k in diff + gone:
                print("    ", "CHANGED" if k in diff else "GONE   ", k[:110])
            bad += len(diff) + len(gone)
        for n in sorted(set(os.path.basename(p) for p in glob.glob(os.path.join(tmp, "old", "*.s"))) -
                        set(os.path.basename(p) for p in glob.glob(os.path.join(tmp, "new", "*.s")))):
            print(f"{n[:-2]:12s} file removed ({len(functions(os.path.join(tmp, 'old', n)))} kernels)")
        return 1 if bad else 0
    finally:
        subprocess.run(["git", "-C", ROOT, "worktree", "remove", "--force", wt])
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
