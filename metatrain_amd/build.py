"""Build libpet_hip.so for gfx950 with hipcc (in-tree, so it travels with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libpet_hip.so")
SOURCES = ["abi.hip", "graph.hip", "nl.hip", "pet_fwd.hip", "pet_bwd.hip", "pet_trr.hip", "pet_attn.hip", "pet_ablk.hip", "pet_ablk_bwd1.hip", "pet_emlp_s.hip", "pet_head_s.hip", "pet_compress_s.hip", "pet_center_s.hip", "pet_comb.hip", "pet_comb_s.hip", "pet_comb_bwd.hip", "pet_comb_bwd_s.hip", "train.hip", "optim.hip", "so.hip", "so_rows_s.hip", "pet_node_s.hip", "soap.hip", "gen.hip", "gen_train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc"]
FLAGS += os.environ.get("PET_HIP_EXTRA_FLAGS", "").split()  # debugging builds, e.g. -DAB_PROFILE (pet_ablk.hip)
# Translation units compiled on their own (no -fgpu-rdc: their device code is generated here, not at the link step) with the
# matrix products in VGPR form. A kernel at one wave per SIMD otherwise gets the AGPR form of every MFMA and pays 16
# v_accvgpr_read for every product tile that vector arithmetic consumes (pet_ablk_bwd1.hip has the numbers). Per file, because
# the option is not a win everywhere (k_comb_p2 loses 5 % with it) and crashes this compiler on some kernels.
VGPR_FORM = {"pet_ablk_bwd1.hip", "pet_comb_bwd.hip"} if "-DAB_PROFILE" not in FLAGS else set()
VGPR_FORM_FLAGS = ["-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form"]
# Translation units whose device code is generated per file, not at the link step (-fgpu-rdc generates it there, over all files, and
# its register allocation of k_comb_bwd_s came out at 256 registers + 4 spilled -- scratch loads inside the ring's vmcnt window --
# where the per-file code generation needs 229 and none)
NO_RDC = {"pet_comb_bwd_s.hip", "so_rows_s.hip"}


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "pet_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    # the flag set is part of an object's identity: a stamp file in the object directory forces a full rebuild when it changes
    # (PET_HIP_EXTRA_FLAGS = -DAB_PROFILE also switches VGPR_FORM off: objects of the other setting must not be linked)
    stamp = os.path.join(objdir, "flags.stamp")
    flag_id = " ".join(FLAGS) + " | " + " ".join(sorted(VGPR_FORM)) + " | " + " ".join(VGPR_FORM_FLAGS) + " | " + " ".join(sorted(NO_RDC))
    if not os.path.exists(stamp) or open(stamp).read() != flag_id:
        have_objects = any(f.endswith(".o") for f in os.listdir(objdir))
        force = force or os.path.exists(stamp) or (have_objects and bool(os.environ.get("PET_HIP_EXTRA_FLAGS")))
        with open(stamp, "w") as fh:
            fh.write(flag_id)
    procs, objs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or hdr_time > os.path.getmtime(o):
            cmd = ["hipcc", *FLAGS, *(VGPR_FORM_FLAGS if src in VGPR_FORM else []), *(["-fno-gpu-rdc"] if src in NO_RDC else []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f"hipcc failed for {failed}")
    if procs or not os.path.exists(LIB):
        # (-fgpu-rdc: the device code is generated at this step, so code-generation switches belong here too)
        cmd = ["hipcc", "--offload-arch=gfx950", "-fgpu-rdc", "-shared", "-fPIC", *os.environ.get("PET_HIP_LINK_FLAGS", "").split(),
               *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


TORCH_LIB = os.path.join(HERE, "lib", "libpet_hip_torch.so")


def build_torch_ops(force: bool = False, verbose: bool = True) -> str:
    """TorchScript-visible wrapper (csrc/torch_ops.cpp): plain C++ against the torch headers, linked to
    libpet_hip.so. g++ only -- there is no device code in it."""
    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(CSRC, "torch_ops.cpp")
    hdr = os.path.join(os.path.dirname(HERE), "include", "pet_hip.h")
    if not (force or _newer(src, TORCH_LIB) or _newer(hdr, TORCH_LIB) or _newer(LIB, TORCH_LIB)):
        return TORCH_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", src, "-o", TORCH_LIB,
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           *[f"-I{p}" for p in ce.include_paths()], "-I/opt/rocm/include",
           f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           f"-L{os.path.dirname(LIB)}", "-lpet_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_torch_ops(force="--force" in sys.argv)
