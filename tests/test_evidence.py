"""The committed measurement record must belong to the committed kernels: every section of
profiles/pmc_traffic.json -- the HBM-traffic figures bench.py quotes as `roofline.traffic` -- carries the hash of the kernel sources
(paif_amd/csrc/*.hip, *.h) it was measured on; it must be the hash of the sources in this tree, so that the driver's bench line
carries no `traffic_note`.  (Re-collect with tools/collect_r06.sh + tools/collect_r06_copy.sh after the last kernel change.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_record_matches_the_kernel_sources():
    import bench

    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    sha = bench.kernel_source_sha16()
    secs = {k: v for k, v in rec.items() if not k.startswith("_")}
    assert {"fusion/f16", "fusion/f32", "fusion/bf16", "fusion_seg/f16"} <= set(secs), sorted(secs)
    for k, v in secs.items():
        assert v.get("_kernel_source_sha16") == sha, (k, v.get("_kernel_source_sha16"), sha)
        assert v.get("_round") == 6, k
    # the default bench line's dominant family (the LDS-DMA 3x3 convs of the fp16 forward) has its traffic in the record
    f16 = secs["fusion/f16"]
    dma = [k for k in f16 if k.startswith("conv3x3_h16_dma<")]
    assert len(dma) >= 4 and all(f16[k]["traffic_bytes"] > 0 for k in dma), dma


def test_parity_report_of_the_default_storage_mode_is_committed():
    import bench

    blk = bench.parity_block("f16")
    assert "note" not in blk or "unreadable" not in blk["note"], blk
    assert blk["clause_argmax_ge_0.999"] is True and blk["clause_miou_within_0.1pt"] is True, blk
    assert blk["samples"] == 32 and blk["clause_argmax_ge_0.999_at_the_lower_end_of_the_interval"] is True, blk
