"""The bench line at N > 1, end to end on a one-GPU box: two ranks folded onto the GPU that exists (BENCH_FOLD_RANKS=1: the harness
exchange stands in for RCCL, which cannot put two ranks on one device), launched exactly as the driver launches N > 1, with the
secondary workloads shrunk (--quick). Checks the contract's keys and the lines VERDICT r3 asked for at N > 1: the split MSM, the
NTT replicas (elements/s summed over the ranks), three Rep3 parties with a GPU each, the placed prover."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
    return json.loads(lines[-1])


def test_two_rank_line_folded_on_one_gpu(gpu):
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                 "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "16", "--quick"], {"BENCH_FOLD_RANKS": "1"})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "result_check", "secondary"):
        assert k in line, k
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["unit"] == "points/s" and line["result_check"] is True
    assert line["config"]["points_total"] == 2 << 16 and "harness all-gather" in line["config"]["split"]         # the exchange that ran is named
    roof = line["roofline"]
    assert roof["bound"] == "valu_int_mad" and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-2)
    assert roof["alu"]["peak_nominal"] == pytest.approx(39.3, abs=0.1) and 0 < roof["alu"]["frac_vs_nominal"] < 1
    # the integer-issue figures and the plan as SCALAR fields of the roofline object (the driver's record keeps only those)
    for k in ("alu_frac", "alu_achieved_Tmad", "alu_peak_Tmad", "alu_frac_vs_nominal", "accum_ms", "c", "windows", "hbm_frac"):
        assert isinstance(roof[k], (int, float)) and roof[k] > 0, k
    assert roof["alu_frac"] == roof["alu"]["frac"] and roof["c"] == roof["msm_params"]["c"]
    # the exchange that ran and the ranks RCCL saw are keys of config, not free text (folded ranks: the harness exchange)
    assert line["config"]["exchange_mode"] == "harness_gloo" and line["config"]["rccl_ranks_seen"] == 0
    sec = line["secondary"]
    assert "error" not in sec, sec
    for wl in ("bn254_g1", "bls12_381_g1", "bls12_381_g2"):
        e = sec[f"msm_{wl}_2p18_strong"]
        assert e["result_check"] is True and e["ranks"] == 2 and e["points_per_rank"] == 1 << 17
    rep = sec["ntt_bn254_2p16_replicas"]
    assert rep["ranks"] == 2 and rep["elements_per_s"] == pytest.approx(2 * (1 << 16) / rep["ms_per_transform_slowest_rank"] * 1e3, rel=1e-6)
    p3 = sec["groth16_rep3_party_per_gpu"]
    assert p3["proofs_equal_plain"] is True and p3["party_devices"] == [0, 0, 0] and p3["three_parties_prove_ms"] > 0
    placed = sec["groth16_prove_synthetic_2p14_placed"]
    for mode in ("by_query", "by_range"):                                     # one prover, its five query MSMs placed on the (folded) GPUs
        assert "error" not in placed[mode], placed[mode]
        assert placed[mode].get("result_check", True) is not False and placed[mode].get("check", True) is not False, placed[mode]
    sp = sec["single_process_split_bn254_g1_2p18"]                            # one thread driving every range: both copy exchanges agree
    assert "error" not in sp and sp["folded_on_device_0"] is True and sp["exchanges_agree"] is True, sp
    assert sp["hipMemcpyPeer_ms"] > 0 and sp["host_copies_ms"] > 0 and "rccl_grouped_skipped" in sp
    # the spin-up runs whatever flags were passed and is recorded; the cold figure stays visible beside the value
    assert line["config"]["spinup_steps"] >= 10 and line["config"]["spinup_ms"] >= 300 and line["value_first_20_steps"] > 0


def test_single_gpu_line_quick(gpu):
    line = _run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--log-n", "16", "--quick", "--no-cpu-baseline"], {})
    assert line["n_gpus"] == 1 and line["result_check"] is True and line["cpu_baseline"] is None
    sec = line["secondary"]
    assert "error" not in sec, sec
    ntt = sec["ntt_bn254_2p16"]
    assert ntt["ms"] > 0 and ntt["ms_first_batch_after_idle"] > 0 and ntt["warm_up_transforms"] >= 120 and ntt["roofline"]["alu"]["frac"] > 0
    assert line["config"]["spinup_steps"] >= 10 and line["config"]["spinup_ms"] >= 300 and line["value_first_20_steps"] > 0
    assert sec["msm_bn254_g1_2p18"]["spinup_steps"] >= 3
    for key in ("msm_bn254_g1_2p20_fixed_base_tables", "msm_bn254_g1_2p18_fixed_base_tables"):    # round 6: one bucket set, 17-bit windows (the library's policy)
        t = sec[key]
        assert "error" not in t and t["result_check"] is True and t["bucket_sets"] == 1 and t["window_bits"] == 17 and t["points_per_s"] > 0, t
    pr = sec["groth16_prove_synthetic_2p14"]
    for mode in ("host_masks", "seeded_device_masks"):                        # BASELINE config 4 through the zero-upstream-edit path, both mask modes
        m = pr["rep3_trait_path"][mode]
        assert m["proofs_equal_plain"] is True and m["three_parties_one_gpu_ms"] > 0 and m["one_party_alone_ms"] > 0, m
    assert pr["rep3_trait_path"]["seeded_device_masks"]["party0_phases_ms"]["mask_draw"] < pr["rep3_trait_path"]["host_masks"]["party0_phases_ms"]["mask_draw"]
    assert pr["prove_ms"] >= pr["prove_ms_min"] and pr["trait_path_ms"] >= pr["trait_path_ms_min"]   # medians, the minimum beside them
    assert pr["closed_form_check"] is True and pr["trait_path_closed_form_check"] is True and pr["trait_path_ms"] > 0 and pr["rep3_proofs_equal_plain"] is True
