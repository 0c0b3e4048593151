#!/usr/bin/env python
"""BASELINE.md §8 from ONE committed bench line (+ the ops tables of the same evidence run): every figure in that section is
printed by this script from profiles/<tag>_bench_n1.json, profiles/<tag>_ops_table*.md and profiles/<tag>_bench_2ranks_one_gpu*.json.

    python tools/baseline_table.py r04 > /tmp/section8.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda name: os.path.join(ROOT, "profiles", name)  # noqa: E731
d = json.loads(open(P(f"{tag}_bench_n1.json")).read().strip().splitlines()[-1])
ex, par, roof, cpu = d["extras"], d["parity"], d["roofline"], d["cpu_baseline"]


def ops_rows(path):
    out = {}
    for line in open(path):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 5 and c[3].replace(".", "").isdigit():
            out[c[1]] = (c[2].strip("`"), float(c[3]), float(c[4]), float(c[5]))
    return out


ops, ops32 = ops_rows(P(f"{tag}_ops_table.md")), ops_rows(P(f"{tag}_ops_table_f32.md"))
n = d["config"]["n_qubits"]
g = d["config"]["gates_per_step"]
L = print
L(f"| config | backend | ms | gates/s (ops/s) | GB/s | % of 8 TB/s | parity |")
L("|---|---|---|---|---|---|---|")
L(f"| **headline**: {g} random single-qubit gates (H / X / Rz), **n = {n}**, f64, {d['steps']} steps | HIP kernels, 1 × MI355X | {d['ms_per_step'] / g:.2f} per gate | "
  f"{d['gates_per_s']:.1f} | **{d['value']:.0f}** | **{100 * d['value'] / 8000:.1f}** | `parity_ok` = {d['parity_ok']}: {par['gates_checked']} gates in {len(par['legs'])} legs on closed "
  f"sub-cubes ({par['rows_checked']:.2e} rows) vs the CPU oracle, max \\|Δ\\| = {par['max_abs_delta']:g} in the IEEE-equal legs, {par['max_abs_delta_1e-12_legs']:.1e} in the 1e-12 legs; "
  f"{par['whole_vector']['compares']} whole-vector compares of 2^{n} amplitudes against a twin on the literal kernel: {par['whole_vector']['amplitudes_not_equal_in_IEEE_legs']} unequal |")
m = d["mixed_circuit"]
L(f"| configs[1] mix (¾ of those + ¼ CNOT), n = {n} | HIP kernels | {m['ms'] / m['gates']:.2f} per gate | {m['gates_per_s']:.1f} | {m['algorithmic_GBps']:.0f} | {100 * m['frac_of_8TBps']:.1f} | same block |")
hs = ex["h_sweep_min_median_GBps"]
L(f"| H on each target qubit, n = {n}: min / median | HIP kernels | {32 * 2**n / hs[0] / 1e6:.2f} / {32 * 2**n / hs[1] / 1e6:.2f} per gate | | {hs[0]:.0f} / {hs[1]:.0f} | {hs[0] / 80:.1f} / {hs[1] / 80:.1f} | |")
c28 = ex["configs1_n28"]
L(f"| configs[1] exactly: n = 28, 256 gates | HIP kernels | {c28['ms_per_step'] / 256:.2f} per gate | {c28['gates_per_s']:.0f} | {c28['GBps']:.0f} | {c28['GBps'] / 80:.1f} | oracle windows at n = 28 |")


def leg(key, label, sub=None, bar="IEEE-equal to gate by gate; oracle + twin at n = 30 (`parity.legs`)"):
    v = ex[key] if sub is None else ex[key][sub]
    cnt = v.get("gates", v.get("ops"))
    L(f"| {label} | HIP kernels + hiprtc | {v['ms']:.1f} ({v['launches']} sweeps, {v['ms'] / v['launches']:.2f} each) | {cnt / v['ms'] * 1e3:.0f} | {v['per_launch_GBps']:.0f} per sweep | "
      f"{v['per_launch_GBps'] / 80:.0f} per sweep | {bar} |")


t12 = "≤ 1e-12 (oracle + twin at n = 30)"
leg("tiled_mode1", f"configs[1] at n = {n} as `tile = 1` sweeps, interpreter kernel", bar="IEEE-equal")
leg("tiled_mode1_jit", "same, segments compiled at run time (`tile_jit`)")
leg("tiled_mode1_jit_relabel", "same + the scheduler relabelling the qubits (`tile_relabel`)")
leg("tiled_mode1_jit_wide", "**wide tiles** (`tile_wide`: 13-bit register-resident tile, seven free positions), circuit order")
leg("tiled_mode1_jit_wide_relabel", "wide tiles + relabelling")
leg("tiled_mode2_jit_fma_relabel", "`tile = 2` + fused multiply-adds + merged diagonal runs + relabelling (11-bit tile)", bar=t12)
if "tiled_mode2_jit_fma_merge_wide" in ex:
    leg("tiled_mode2_jit_fma_merge_wide", "**`tile = 2` + fma + merged diagonal runs over wide tiles**", bar=t12)
leg("fused_k5", "dense fusion (`fuse = 5`): one sweep per fused gate", bar=t12)
for key, label in (("configs2_qft_n%d" % n, "configs[2] QFT (480 ops)"), ("configs3_clifford_t_n%d" % n, "configs[3] Clifford+T on one GPU (256 ops)"),
                   ("configs4_grover_iteration_n%d" % n, "configs[4] one Grover iteration (182 ops)"), ("configs4_grover_dense_k3_n%d" % n, "configs[4] Grover, dense-k3 variant (170 ops)")):
    v = ex[key]
    L(f"| {label}, n = {n}: gate by gate | HIP kernels | {v['ms']:.0f} | {v['ops'] / v['ms'] * 1e3:.0f} | {v['algorithmic_GBps']:.0f} | {v['algorithmic_GBps'] / 80:.1f} | oracle + twin at n = 30 |")
    for sub in ("tile1", "tile1_jit", "tile1_jit_relabel", "tile1_jit_wide", "tile1_jit_wide_relabel", "tile2_jit_fma_merge", "tile2_jit_fma_merge_relabel",
                "tile2_jit_fma_merge_wide", "tile2_jit_fma_merge_wide_relabel"):
        if sub in v:
            leg(key, f"  … `{sub}`", sub, bar=t12 if "tile2" in sub else "IEEE-equal")
f32 = ex["complex64_n%d" % n]
L(f"| Complex<f32>, the headline circuit gate by gate, n = {n} (8 GiB) | HIP kernels | {f32['ms'] / f32['gates']:.2f} per gate | {f32['gates_per_s']:.0f} | {f32['algorithmic_GBps']:.0f} | "
  f"{100 * f32['frac_of_8TBps']:.1f} | f32 oracle windows at n = 30 |")
L(f"| Complex<f32>, configs[1] as `tile = 1` compiled + relabelled | HIP kernels | {f32['mixed_tile1_jit_relabel']['ms']:.1f} | {f32['mixed_tile1_jit_relabel']['gates_per_s']:.0f} | | | 1e-5 |")
if "mixed_tile1_jit_wide" in f32:
    L(f"| Complex<f32>, configs[1] as `tile = 1` over wide tiles | HIP kernels | {f32['mixed_tile1_jit_wide']['ms']:.1f} | {f32['mixed_tile1_jit_wide']['gates_per_s']:.0f} | | | 1e-5 (bit-identical to the narrow f32 sweeps) |")
for name in ("dense k=2 (VALU regs)", "controlled dense k=2, low targets, control above the rows", "2-controlled dense k=3, one low target", "dense k=3 (MFMA f64)", "dense k=4 (MFMA f64)",
             "dense k=4 high bits (MFMA f64)", "dense k=4 (VALU regs)", "dense k=5 (MFMA f64)", "dense k=5 high bits (MFMA f64)"):
    if name in ops:
        k, ms, gb, pc = ops[name]
        L(f"| {name} | `{k}` | {ms:.2f} | | {gb:.0f} | {pc:.1f} | `tests/test_parity_gpu.py` |")
for name in ops:
    if name.startswith("dense k=6 (MFMA") or name.startswith("dense k=7") or name.startswith("dense k=8") or name.startswith("dense k=9") or name.startswith("dense k=10"):
        k, ms, gb, pc = ops[name]
        kk = int(name.split("=")[1].split()[0])
        tf = 8.0 * 2**kk * 2**n / (ms * 1e-3) / 1e12
        L(f"| {name} | `{k}` | {ms:.1f} | | {tf:.1f} TFLOP/s | **{100 * tf / 78.6:.0f} % of the 78.6 TFLOP/s f64 matrix peak** | 1e-12 vs oracle |")
for name in ("sparse k=4, 2 entries per row (in place)", "sparse k=4, 2 entries per row (one group per lane)", "sparse k=5, 2 entries per row (in place)",
             "sparse k=5, 2 entries per row (one group per lane)", "sparse k=16 identity, one entry per row (state_bench.rs:380-393 shape)",
             "sparse k=8 permutation x phase, scattered bits", "sparse k=8 permutation x phase, scattered bits (out-of-place gather)", "sparse k=6, 2 entries per row",
             "sparse k=6, 2 entries per row (out-of-place gather)", "sparse k=7, 4 entries per row, two positions in the wave row", "controlled sparse k=6, 2 entries per row"):
    if name in ops:
        k, ms, gb, pc = ops[name]
        b = ops32.get(name)
        L(f"| `SparseMatrix`: {name.replace('sparse ', '', 1)} | `{k}` | {ms:.2f} | | {gb:.0f} | {pc:.1f}" + (f" (f32: `{b[0]}` {b[3]:.1f})" if b else "") + " | bit-equal to the oracle |")
for name in ("norm_sqr", "measure_probs k=1", "measure_probs k=3", "measure_probs k=12 top bits", "measure_probs k=12 mixed bits", "measure_probs k=16", "soft_measure (2 passes)",
             "soft_measure (one launch: chunk sums + last block's walk; measured alternative)"):
    a, b = ops.get(name), ops32.get(name)
    if a and b:
        L(f"| {name}: f64 / f32 | reduction | {a[1]:.2f} / {b[1]:.2f} | | {a[2]:.0f} / {b[2]:.0f} | {a[3]:.1f} / {b[3]:.1f} | ≤ 1e-13 |")
L(f"| first {cpu['sample'].split()[1]} gates of the headline circuit at **n = 28** | CPU oracle ('{cpu['kind']}', gcc -O3 -fopenmp) | {cpu['ms_per_gate']:.0f} per gate | {cpu['gates_per_s']:.1f} | **{cpu['value']:.1f}** | — | "
  f"{cpu['cores']} threads = the cgroup CPU quota ({cpu['cores_usable']} usable by affinity); {cpu['ns_per_row_per_thread']:.1f} ns per row per thread, thread scaling {100 * cpu['thread_scaling_efficiency']:.0f} % |")
L("")
L(f"Dominant kernel (`roofline`): `{roof['kernel']}`, {roof['launches']} launches, {roof['avg_launch_ms']:.3f} ms average (HIP events) = {roof['achieved']:.0f} GB/s = "
  f"**{100 * roof['frac']:.1f} %**; HBM traffic per launch (`roofline.traffic`, {'STALE: ' if roof.get('traffic_stale') else ''}{roof['traffic_source'].split('(')[0].strip()}): "
  f"{roof['traffic']:.4e} B against {roof['algorithmic_bytes_per_launch']:.4e} algorithmic = {roof['traffic'] / roof['algorithmic_bytes_per_launch']:.4f}×.")

import csv
import glob

rows = [r for r in csv.DictReader(open(P(f"{tag}_kernel_stats.csv"))) if "k_tile_passes" in r["Name"]]
calls = sum(int(r["Calls"]) for r in rows)
avg = sum(int(r["TotalDurationNs"]) for r in rows) / calls / 1e6
L(f"rocprofv3 `--kernel-trace --stats` of the same command (`profiles/{tag}_kernel_stats.md`): `k_tile_passes` {calls} calls, {avg:.3f} ms average.")
L("")
if os.path.exists(P(f"{tag}_bench_n33.json")):
    b = json.loads(open(P(f"{tag}_bench_n33.json")).read().strip().splitlines()[-1])
    L(f"Headline generator at **n = 33** (128 GiB, eight times the headline state) on ONE GPU, in place, {b['config']['gates_per_step']} gates × {b['steps']} steps: "
      f"{b['ms_per_step'] / b['config']['gates_per_step']:.1f} ms per gate = **{b['value']:.0f} GB/s = {b['value'] / 80:.1f} %**, norm {b['norm_sqr_after']:.15f} "
      f"(`profiles/{tag}_bench_n33.json`; oracle windows at n = 33: `tests/test_parity_gpu.py::test_full_size_oracle_windows_n33`).")
    L("")
for f in sorted(glob.glob(P(f"{tag}_bench_2ranks_one_gpu*.json"))):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    pr = b["parity"]
    L(f"Two ranks on ONE GPU (`{os.path.basename(f)}`, host-staged transport: plumbing and parity evidence, not a throughput figure): n = {b['config']['n_qubits']}, "
      f"`parity_ok` = {b['parity_ok']}, parity at n = {pr.get('n')}: {pr.get('rows_checked', 0):.3g} rows, max |Δ| = {pr.get('max_abs_delta')}.")
    for k, v in b.get("extras", {}).items():
        if isinstance(v, dict) and "comm_over_reps" in v:
            c = v["comm_over_reps"]
            L(f"  * `{k}`: {c['remaps']} remaps, {c['pack_sweeps']} gathers as sweeps of their own, **{c['packs_folded']} folded into the preceding tile sweep**, {c['packs_via_permute_bits']} through `k_permute_bits`"
              + (f"; exchange overlapped with the sweep before it in {c['remaps_overlapped']} remaps ({c['remaps_overlapped_after']} also with the sweep after), {c['slices_overlapped']} slices" if c.get("remaps_overlapped") is not None and "overlap" in k else ""))


# ---- r5: who compiled what, programs, and the reference's own bench shapes (printed after the main table) -------------------------------
if "jit" in ex:
    j = ex["jit"]
    L("")
    L(f"Run-time compiler over the whole bench process (`extras.jit`): {j['kernels_resident_total']} segment kernels made resident, {j['compiled']} compiled "
      f"({j['compiled_by_helpers']} of them in {j['helper_processes']} helper processes, at most {j['procs']} side by side), {j['disk_hits']} loaded from the disk cache, "
      f"{j['compile_ms'] / 1e3:.1f} s of wall time compiling in all, {j['disk_load_ms']:.1f} ms reading code objects.")
    for key in ("tiled_mode1_jit",):
        if "compile_ms_once" in ex.get(key, {}):
            L(f"`{key}`: {ex[key]['segments_compiled']} segments, `compile_ms_once` = {ex[key]['compile_ms_once']:.0f} ms.")
    sp = P(f"{tag}_bench_n1_second_process.json")
    if os.path.exists(sp):
        b2 = json.loads(open(sp).read().strip().splitlines()[-1])
        j2 = b2["extras"]["jit"]
        L(f"A SECOND process of the same command on the same box (`{os.path.basename(sp)}`): {j2['compiled']} compiled, {j2['disk_hits']} loaded from disk in {j2['disk_load_ms']:.1f} ms; "
          f"`tiled_mode1_jit.compile_ms_once` = {b2['extras']['tiled_mode1_jit']['compile_ms_once']:.1f} ms.")
if "program_tile_auto" in ex and "ms" in ex["program_tile_auto"]:
    pa = ex["program_tile_auto"]
    L(f"A program created on a `tile` = 1 state (`tile_auto`): {pa['ms']:.1f} ms per replay of configs[1] ({pa['gates_per_s']:.0f} gates/s), hipGraph = {pa['is_graph']}, created in {pa['create_s_once']:.2f} s (compilation included).")
if "reference_bench_shapes" in ex:
    L("")
    L("### The reference's own benches, at the reference's sizes (`extras.reference_bench_shapes`; microseconds per `apply_op`)")
    L("")
    L("| reference bench | n | element | eager (one C-ABI call per op) | hipGraph program of 64 | tiled program of 64 | CPU restatement (`apply_op`, accumulate): all granted threads / one thread | algorithmic bytes per op |")
    L("|---|---|---|---|---|---|---|---|")
    for name, r in ex["reference_bench_shapes"].items():
        f = lambda k: (f"{r[k]:.2f}" if k in r else "—")  # noqa: E731
        L(f"| `{name}` | {r['n']} | {r['dtype']} | {f('eager_us_per_op')} | {f('hipgraph_program_us_per_op')} | {f('tiled_program_us_per_op')} | "
          f"{f('cpu_restatement_us_per_op')} ({r.get('cpu_threads', '?')} threads) / {f('cpu_restatement_one_thread_us_per_op')} | {r['algorithmic_bytes_per_op']:.3g} |")
