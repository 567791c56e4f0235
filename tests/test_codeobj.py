"""Static checks on the compiled gfx950 code objects inside libhtf_amd.so (no GPU needed):
the hot kernels must not touch scratch memory.  A by-value parameter struct indexed at run
time is silently moved to private memory by the compiler -- that doubled the LJ evaluator's
time once (every wave spilling its PotParams copy) without failing a single parity test."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hoomd_tf_amd", "libhtf_amd.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _kernel_metadata(tmp_path):
    data = open(LIB, "rb").read()
    meta = {}
    for i, m in enumerate(re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)):
        p = m.start()
        (n,) = struct.unpack_from("<Q", data, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" not in triple or size == 0:
                continue
            f = tmp_path / ("co%d.elf" % i)
            f.write_bytes(data[p + off:p + off + size])
            txt = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
            for blk in txt.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                              for k in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
                                        "vgpr_count", "group_segment_fixed_size")}
    return meta


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="library or llvm-readelf missing")
def test_hot_kernels_use_no_scratch(tmp_path):
    meta = _kernel_metadata(tmp_path)
    assert len(meta) > 50
    hot = [n for n in meta if any(t in n for t in (
        "build_pair_vectors_kernel", "eval_pair_kernel", "eval_pair2_kernel", "train_pair_kernel",
        "mlp_grad_mfma_kernel", "mlp_grad_kernel", "rdf_hist", "nve_step_kernel", "fused_forces_rows2_kernel",
        "fused_forces2_kernel", "build_nlist_kernel", "cell_ranges_kernel", "cell_order_kernel", "topk_mlp_kernel",
        "topk_values_kernel", "positions_radial_kernel", "commit_rebuild_kernel",
        # the kernels bench.py times: the one-kernel LJ / WCA step (fp32 and fp64 wire, 2-4 rows per wave), the C4 sweep and the
        # pair-MLP training sweep on the fp16 pipeline (one wave per SIMD, 450 registers: at the edge of the file)
        "fused_forces_tails_kernel", "fused_forces2_tails_kernel", "mlp_grad_tr16_kernel"))]
    # (polynomial + virial + fp64 positions spills a few SGPRs: a rare combination, left alone)
    hot += [n for n in meta if "fused_forces_kernel" in n
            and not re.match(r"_ZN3htf19fused_forces_kernelILi3ELb1ELb[01]EdEE", n)]
    hot += [n for n in meta if "pair_mlp_kernel" in n]  # fp32, bf16 and split images
    assert len(hot) > 40
    bad = {n: meta[n] for n in hot if meta[n]["private_segment_fixed_size"] or meta[n]["vgpr_spill_count"]}
    assert not bad, bad
    # the matrix-core training kernels need their 151-157 KB of LDS to fit the CU's 160 KB
    for n in meta:
        if "mlp_grad_mfma_kernel" in n or "mlp_grad_tr16_kernel" in n:
            assert meta[n]["group_segment_fixed_size"] <= 160 * 1024
