"""What the compiler made of the step kernels, read from the code objects inside the built library (CPU test: no GPU needed).

Round 3 found the fp32 half-step kernels 5x slower than the fp64 ones: a copied 32-byte draw record that the compiler could
not take apart into registers stayed a private array, the back end moved it to LDS ("promote alloca to LDS"), and a kernel
with such an array reads the workgroup size from the dispatch packet -- host memory -- at the start of every wavefront:
13-28 us per launch (profiles/r03_fp32_half_step_anomaly.txt).  Nothing in the sources says "LDS" or "scratch" when that
happens, so the metadata is checked here: the kernels of the step path (stretch_*, de_update*, calc_logp) may use the
dynamic LDS their launchers ask for, but no static LDS and no scratch memory."""
import os
import re
import subprocess

from mcmcpp_amd import capi

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# (one instantiation nobody has asked for yet: fp64 dense, D = 1024 through the plain full-step kernel, 144 bytes of spills)
KNOWN_SCRATCH = {"stretch_full_step_kernel<double, DenseGaussianFn<double>, 16, 64"}


def _kernels_of_library(tmp_path):
    lib = capi.build_library()
    fat = tmp_path / "fatbin.bin"
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=%s" % fat, lib, str(tmp_path / "copy.so")])
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)] + [len(data)]
    kernels = []
    for k in range(len(starts) - 1):
        bundle = tmp_path / ("bundle%d.bin" % k)
        bundle.write_bytes(data[starts[k]:starts[k + 1]])
        co = tmp_path / ("bundle%d.co" % k)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=%s" % bundle,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=%s" % co])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
        for block in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            kernels.append({"symbol": name,
                            "lds": int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1)),
                            "scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)),
                            "vgprs": int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1))})
    names = subprocess.run(["c++filt"], input="\n".join(k["symbol"] for k in kernels), capture_output=True, text=True, check=True).stdout.split("\n")
    for k, n in zip(kernels, names):
        k["name"] = n.replace("void mcmcpp::", "").replace("mcmcpp::", "", 1) if n else k["symbol"]
    return kernels


def test_step_kernels_carry_no_promoted_private_arrays_and_no_scratch(tmp_path):
    kernels = _kernels_of_library(tmp_path)
    step = [k for k in kernels if re.match(r"(void )?(mcmcpp::)?(stretch_(half|full)_step|de_update|calc_logp)", k["name"])]
    assert len(step) > 400, "the library's step kernels were not found in its code objects (%d kernels seen)" % len(kernels)
    assert any("<float" in k["name"] for k in step) and any("<double" in k["name"] for k in step)
    offenders = [(k["name"][:110], k["lds"], k["scratch"]) for k in step
                 if (k["lds"] != 0 or k["scratch"] != 0) and not any(k["name"].startswith(p) or p in k["name"] for p in KNOWN_SCRATCH)]
    assert not offenders, "static LDS / scratch in step kernels (name, LDS bytes, scratch bytes): %r" % offenders[:6]
    # the headline kernel keeps its one-workgroup-per-CU budget (8 wavefronts of 182 registers)
    head = [k for k in step if k["name"].startswith("stretch_full_step_mfma_kernel<double")]
    assert head and all(k["vgprs"] <= 192 for k in head), [(k["name"][:60], k["vgprs"]) for k in head]
    # the 16-walker matrix-core half-step kernel: three wavefronts per SIMD with the next draws in the gather's shadow, four
    # (LATE, the last template argument) with the draws behind the accept -- the point of the round-3 rework
    mc16 = [k for k in step if re.match(r"stretch_half_step_mfma_kernel<double, DenseGaussianFn<double>, 2, 16, 4, false, (true|false), (true|false)>", k["name"])]
    assert len(mc16) == 4, [k["name"][:100] for k in mc16]
    for k in mc16:
        late = k["name"].split(">(")[0].endswith("true")
        assert k["vgprs"] <= (128 if late else 168), (k["name"][:100], k["vgprs"])
