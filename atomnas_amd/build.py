"""Builds libatomnas_hip.so (gfx950) in-tree with hipcc.  `python -m atomnas_amd.build` or `build_library()`.

The library is a plain C-ABI shared object (include/atomnas_hip.h); it does not link against torch.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libatomnas_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


# The depthwise kernels issue asynchronous loads from inline asm and wait for them by hand (csrc/dwconv_cw.hip cw_row_issue): whether the
# compiler leaves those registers alone between issue and wait was checked on the ISA of this toolchain (tools/check_asm_waits.py,
# profiles/r04_asm_wait_check.txt).  Another hipcc gets a warning: re-run the check before trusting its build.
HIPCC_VALIDATED = "HIP version: 7.2"
_warned = [False]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the AtomNAS HIP kernels cannot be built")
    if not _warned[0]:
        _warned[0] = True
        try:
            ver = subprocess.run([exe, "--version"], capture_output=True, text=True).stdout
            if HIPCC_VALIDATED not in ver:
                sys.stderr.write("atomnas_amd.build: hipcc is not the validated %s toolchain (%s): run `python tools/check_asm_waits.py "
                                 "atomnas_amd/csrc/dwconv_cw.hip atomnas_amd/csrc/dwconv.hip` on its output\n" % (HIPCC_VALIDATED, ver.splitlines()[0] if ver else "?"))
        except OSError:
            pass
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


def _deps(path, seen=None):
    """the quoted includes of a source, transitively (paths relative to the including file)"""
    seen = [] if seen is None else seen
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('#include "'):
                inc = os.path.normpath(os.path.join(os.path.dirname(path), line.split('"')[1]))
                if inc not in seen and os.path.exists(inc):
                    seen.append(inc)
                    _deps(inc, seen)
    return seen


def _digest(path):
    h = hashlib.sha1()
    for p in [path] + sorted(_deps(path)):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def sources_digest():
    """sha1 over every kernel source, the headers of csrc/ and the compiler flags: identifies the library a measurement belongs to
    (profiles/*_pmc_*.json carry it as `lib_src_sha`; bench.py drops PMC-derived figures taken on another build)"""
    h = hashlib.sha1()
    for p in sources() + headers():
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    stamp = obj + ".sha1"
    dig = _digest(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


# sources whose inline-asm loads / counted LDS-DMA waits tools/check_asm_waits.py verifies on the ISA: their device assembly is kept
# next to the objects (csrc/build/*.s, stamped with the same digest) so that the check in the CPU test suite costs a parse, not a compile
ISA_CHECKED = ("dwconv_cw.hip", "dwconv.hip", "dwconv_mm.hip", "dwconv_mm2.hip", "pwconv.hip", "pwconv_tn.hip", "xbwd.hip")


def assemble(src):
    """device assembly of one source (cached by digest) -> path of the .s file"""
    out = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".s")
    stamp = out + ".sha1"
    dig = _digest(src)
    if not (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig):
        os.makedirs(OBJ_DIR, exist_ok=True)
        r = subprocess.run([_hipcc()] + FLAGS + ["--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed for %s:\n%s" % (src, r.stderr[-4000:]))
        with open(stamp, "w") as f:
            f.write(dig)
    return out


def build_library(verbose=False, with_asm=None):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    if with_asm is None:
        with_asm = os.environ.get("ATOMNAS_BUILD_ASM", "1") != "0"
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(12, len(srcs) + len(ISA_CHECKED))) as ex:
        asm = [ex.submit(assemble, s) for s in srcs if with_asm and os.path.basename(s) in ISA_CHECKED]
        results = list(ex.map(_compile, srcs))
        for a in asm:
            a.result()
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIB_PATH):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built" if rebuilt else "up to date", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build_library(verbose=True)
    sys.exit(0)
