#!/usr/bin/env python3
"""Build ceiling-probe copies of the library: conv_mfma.hip recompiled with -DY6_PIPE_PROBE=n (WRONG RESULTS,
timing only - see the comment at kPipeProbe), linked with the regular objects of the other sources into
tools/_build/libyolov6_hip_probe<n>.so.  Use with  Y6_LIB_PATH=tools/_build/libyolov6_hip_probe5.so python tools/conv_bench.py ..."""
import concurrent.futures as cf, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov6_amd", "csrc"))
import build as B  # noqa: E402

B.build(verbose=False)
others = [o for o in glob.glob(os.path.join(B.OBJ_DIR, "*.o")) if not os.path.basename(o).startswith("conv_mfma.")]
out = os.path.join(ROOT, "tools", "_build")
os.makedirs(out, exist_ok=True)
cc = B.hipcc()


def one(n):
    obj = os.path.join(out, f"conv_mfma_probe{n}.o")
    lib = os.path.join(out, f"libyolov6_hip_probe{n}.so")
    subprocess.run([cc] + B.COMMON + B.SOURCES["conv_mfma.hip"] + [f"-DY6_PIPE_PROBE={n}", "-c",
                    os.path.join(B.HERE, "conv_mfma.hip"), "-o", obj], check=True, capture_output=True)
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + sorted(others), check=True)
    os.remove(obj)
    return lib


def one_dma(n):
    """conv_dma.hip rebuilt with -DY6_DMA_PROBE=n -> tools/_build/libyolov6_hip_dmaprobe<n>.so"""
    oth = [o for o in glob.glob(os.path.join(B.OBJ_DIR, "*.o")) if not os.path.basename(o).startswith("conv_dma.")]
    obj = os.path.join(out, f"conv_dma_probe{n}.o")
    lib = os.path.join(out, f"libyolov6_hip_dmaprobe{n}.so")
    subprocess.run([cc] + B.COMMON + B.SOURCES["conv_dma.hip"] + [f"-DY6_DMA_PROBE={n}", "-c",
                    os.path.join(B.HERE, "conv_dma.hip"), "-o", obj], check=True, capture_output=True)
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + sorted(oth), check=True)
    os.remove(obj)
    return lib


if len(sys.argv) > 1 and sys.argv[1] in ("--dma-pixmajor", "--dma-planar"):
    # A/B of the halo image of the fp16 stride-1 16-channel-chunk kernels: pixel-major swizzled (the default build since round 3)
    # against the two-plane image (the default of rounds 1-2)
    planar = sys.argv[1] == "--dma-planar"
    oth = [o for o in glob.glob(os.path.join(B.OBJ_DIR, "*.o")) if not os.path.basename(o).startswith("conv_dma.")]
    obj = os.path.join(out, "conv_dma_layout.o")
    lib = os.path.join(out, "libyolov6_hip_dmaplanar.so" if planar else "libyolov6_hip_dmapixmajor.so")
    subprocess.run([cc] + B.COMMON + B.SOURCES["conv_dma.hip"] + ["-DY6_DMA_PLANAR16=" + ("1" if planar else "0"), "-c", os.path.join(B.HERE, "conv_dma.hip"), "-o", obj],
                   check=True, capture_output=True)
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + sorted(oth), check=True)
    os.remove(obj)
    print("built", lib)
    sys.exit(0)

def one_wreg(n):
    """conv_wreg.hip rebuilt with -DY6_WREG_PROBE=n -> tools/_build/libyolov6_hip_wregprobe<n>.so"""
    oth = [o for o in glob.glob(os.path.join(B.OBJ_DIR, "*.o")) if not os.path.basename(o).startswith("conv_wreg.")]
    obj = os.path.join(out, f"conv_wreg_probe{n}.o")
    lib = os.path.join(out, f"libyolov6_hip_wregprobe{n}.so")
    subprocess.run([cc] + B.COMMON + B.SOURCES["conv_wreg.hip"] + [f"-DY6_WREG_PROBE={n}", "-c",
                    os.path.join(B.HERE, "conv_wreg.hip"), "-o", obj], check=True, capture_output=True)
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + sorted(oth), check=True)
    os.remove(obj)
    return lib


if len(sys.argv) > 1 and sys.argv[1] == "--wreg":
    probes = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 7]
    with cf.ThreadPoolExecutor(max_workers=min(len(probes), 8)) as ex:
        for lib in ex.map(one_wreg, probes):
            print("built", lib)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--dma":
    probes = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6]
    with cf.ThreadPoolExecutor(max_workers=len(probes)) as ex:
        for lib in ex.map(one_dma, probes):
            print("built", lib)
    sys.exit(0)

probes = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 6]
with cf.ThreadPoolExecutor(max_workers=len(probes)) as ex:
    for lib in ex.map(one, probes):
        print("built", lib)
