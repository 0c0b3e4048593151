#!/usr/bin/env python
"""A round's section of BASELINE.md from ONE committed bench run: every figure is printed by this script from
profiles/<tag>_bench_line.json (bench.py's contract line), profiles/<tag>_bench_detail.json (its side file) and
profiles/<tag>_ops_table*.md (tools/bench_ops.py of the same evidence run).

    python tools/baseline_table.py r06 > /tmp/section.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(ROOT, "profiles", name)  # noqa: E731
d = json.loads(open(P(f"{tag}_bench_line.json")).read().strip().splitlines()[-1])
det = json.load(open(P(f"{tag}_bench_detail.json")))
ex, par, roof, cpu = det["extras"], det["parity"], d["roofline"], d["cpu_baseline"]
n, g = d["config"]["n_qubits"], d["config"]["gates_per_step"]
L = print


def ops_rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 5 and c[3].replace(".", "").isdigit():
            out[c[1]] = (c[2].strip("`"), float(c[3]), float(c[4]), float(c[5]))
    return out


L("| config | backend | ms | gates/s (ops/s) | GB/s | % of 8 TB/s | parity |")
L("|---|---|---|---|---|---|---|")
L(f"| **headline**: {g} random single-qubit gates (H / X / Rz), **n = {n}**, f64, {d['steps']} steps | HIP kernels, 1 × MI355X | {d['ms_per_step'] / g:.2f} per gate | "
  f"{d['gates_per_s']:.1f} | **{d['value']:.0f}** | **{100 * d['value'] / 8000:.1f}** | `parity_ok` = {d['parity_ok']}: {par['gates_checked']} gates in {par['legs'] if isinstance(par['legs'], int) else len(par['legs'])} legs "
  f"on closed sub-cubes ({par['rows_checked']:.2e} rows) vs the CPU oracle, max \\|Δ\\| = {par['max_abs_delta_IEEE_legs']:g} in the IEEE-equal legs, {par['max_abs_delta_1e-12_legs']:.1e} in the "
  f"1e-12 legs; {par['whole_vector_compares']} whole-vector compares of 2^{n} amplitudes against a twin on the literal kernel: {par['whole_vector_amplitudes_not_equal_IEEE_legs']} unequal |")
L(f"| dominant kernel `{roof['kernel']}` | HIP events inside the timed region | {roof['avg_launch_ms']:.3f} per launch ({roof['launches']} launches) | | {roof['achieved']:.0f} | "
  f"**{100 * roof['frac']:.1f}** | traffic {roof['traffic'] / roof['algorithmic_bytes_per_launch']:.4f} × algorithmic (PMC) |" if roof.get("traffic") else "")
m = det.get("mixed_circuit")
if m:
    L(f"| configs[1] mix (¾ of those + ¼ CNOT), n = {n}, one launch per gate | HIP kernels | {m['ms'] / m['gates']:.2f} per gate | {m['gates_per_s']:.1f} | {m['algorithmic_GBps']:.0f} | "
      f"{100 * m['frac_of_8TBps']:.1f} | core parity block |")
if "h_sweep_min_median_GBps" in ex:
    hs = ex["h_sweep_min_median_GBps"]
    L(f"| H on each target qubit, n = {n}: min / median | HIP kernels | {32 * 2**n / hs[0] / 1e6:.2f} / {32 * 2**n / hs[1] / 1e6:.2f} per gate | | {hs[0]:.0f} / {hs[1]:.0f} | {hs[0] / 80:.1f} / {hs[1] / 80:.1f} | |")
if "configs1_n28" in ex:
    c28 = ex["configs1_n28"]
    L(f"| configs[1] exactly: n = 28, 256 gates | HIP kernels | {c28['ms_per_step'] / 256:.2f} per gate | {c28['gates_per_s']:.0f} | {c28['GBps']:.0f} | {c28['GBps'] / 80:.1f} | oracle windows at n = 28 (GPU suite) |")


def leg(v, label):
    if not v or "ms" not in v:
        return
    cnt = v.get("gates", v.get("ops"))
    L(f"| {label} | HIP kernels{' + hiprtc' if v.get('options', {}).get('tile_jit') else ''} | {v['ms']:.1f} ({v['launches']} sweeps, {v['ms'] / v['launches']:.2f} each) | {cnt / v['ms'] * 1e3:.0f} | "
      f"{v['per_launch_GBps']:.0f} per sweep | {v['per_launch_GBps'] / 80:.0f} per sweep | {v.get('bar', 'IEEE-equal')}: checked on its slice at n = {n} first |")


b = ex.get("builder_one_shot", {})
if b:
    sp = b.get("second_process", {})
    L(f"| **what a `calculate_state` caller gets**: HipBuilder's run loop on configs[1], n = {n} (one `apply_ops`, tile = 1 + relabel) | first process, cold cache: {b.get('first_call_took')} | "
      f"{b.get('first_call_ms', float('nan')):.1f} | | | | IEEE-equal (checked first) |")
    if "ms" in sp:
        L(f"| same call in a SECOND process (tools/builder_one_shot.py) | {'compiled wide sweeps from the disk cache' if sp.get('compiled_sweeps') else 'interpreter'} | **{sp['ms']:.1f}** | "
          f"{sp['gates'] / sp['ms'] * 1e3:.0f} | | | disk hits {sp['jit'].get('disk_hits')}, compiled {sp['jit'].get('compiled')} |")
t = ex.get("tiled", {})
leg(t.get("tile1_jit"), f"configs[1] at n = {n}, segments compiled at run time (`tile_jit`)")
leg(t.get("tile1_jit_wide"), "**wide tiles** (`tile_wide`), circuit order")
leg(t.get("tile1_jit_wide_relabel"), "wide tiles + relabelling")
pa = ex.get("program_tile_auto")
if pa and "ms" in pa:
    L(f"| a PROGRAM created on a `tile` = 1 state (compiled at creation, one hipGraph) | program | {pa['ms']:.1f} | {pa['gates_per_s']:.0f} | | | graph: {pa['is_graph']} |")
for key, label in (("configs2_qft", "QFT (480 ops)"), ("configs3_clifford_t", "Clifford+T (256 gates)"), ("configs4_grover_iteration", "Grover iteration"),
                   ("configs4_grover_dense_k3", "Grover iteration, dense-k3 variant")):
    k = f"{key}_n{n}"
    if k not in ex:
        continue
    v = ex[k]
    L(f"| **{label}**, n = {n}, one launch per op | HIP kernels | {v['ms']:.1f} | {v['ops'] / v['ms'] * 1e3:.0f} | {v['algorithmic_GBps']:.0f} | {v['algorithmic_GBps'] / 80:.1f} | |")
    for sub, sl in (("tile1", "interpreter sweeps"), ("tile1_jit", "compiled sweeps"), ("tile1_jit_wide", "compiled wide sweeps"), ("tile1_jit_wide_relabel", "wide + relabelled")):
        leg(v.get(sub), f"… {sl}")
tol = ex.get("tolerance_modes_1e-12", {})
for k, v in tol.items():
    leg(v, f"1e-12 mode `{k}`")
f32 = ex.get(f"complex64_n{n}")
if f32:
    L(f"| headline circuit on a Complex<f32> state, n = {n} | HIP kernels | {f32['ms'] / f32['gates']:.2f} per gate | {f32['gates_per_s']:.0f} | {f32['algorithmic_GBps']:.0f} | {100 * f32['frac_of_8TBps']:.1f} | |")
    leg(f32.get("mixed_tile1_jit_wide"), "configs[1] mix, Complex<f32>, compiled wide sweeps")
L(f"| CPU baseline ({cpu['kind']}): {cpu['sample']} | C restatement, {cpu['cores']} threads | {cpu['ms_per_gate']:.0f} per gate | {cpu['gates_per_s']:.2f} | **{cpu['value']:.1f}** | {cpu['value'] / 80:.2f} | it IS the checker |")
L("")
L("Reference bench shapes (`qip/benches/state_bench.rs`, `qip-iterators/benches/matmul_bench.rs`), µs per op on a resident state; the op is PREPARED once (a program), as the reference builds it once outside its loop:")
L("")
L("| shape | n | prepared, one call per op | 64-op hipGraph (graph?) | tiled program | one-shot (pack + upload + launch) | CPU restatement (threads) | CPU, one thread |")
L("|---|---|---|---|---|---|---|---|")
for name, r in ex.get("reference_bench_shapes", {}).items():
    L(f"| {name} | {r['n']} ({r['dtype']}) | {r.get('eager_us_per_op', float('nan')):.2f} | {r.get('hipgraph_program_us_per_op', float('nan')):.2f} ({r.get('hipgraph_program_is_graph')}) | "
      f"{r.get('tiled_program_us_per_op', float('nan')):.2f} | {r.get('one_shot_us_per_op', float('nan')):.1f} | {r.get('cpu_restatement_us_per_op', float('nan')):.1f} ({r.get('cpu_threads')}) | "
      f"{r.get('cpu_restatement_one_thread_us_per_op', float('nan')):.1f} |")
gp = ex.get("generic_p_real_vectors")
if gp:
    L("")
    L("`apply_op<P>` for a REAL `P` on device slices (`qip_hip_apply_op_device`; the reference's f64 bench shapes, `qip-iterators/benches/matmul_bench.rs`; op built once, "
      "algorithmic bytes = `sizeof(P)` × 3 × 2^n for an accumulating call; more shapes: `profiles/" + tag + "_real_p.md`):")
    L("")
    L("| n | P | op | reference bench | µs per call | GB/s | % of 8 TB/s | vs the oracle (every row) |")
    L("|---|---|---|---|---|---|---|---|")
    for r in gp:
        L(f"| {r['n']} | {r['P']} | {r['op']} | `{r['ref']}` | {r['us_per_call']:.1f} | {r['algorithmic_GBps']:.0f} | {100 * r['frac_of_8TBps']:.1f} | "
          f"{'bit-equal' if r.get('bit_equal_to_oracle') else 'not checked' if 'bit_equal_to_oracle' not in r else 'DIFFERS'} |")
ops = ops_rows(P(f"{tag}_ops_table.md"))
if ops:
    L("")
    L(f"Per-op table: `profiles/{tag}_ops_table.md` ({len(ops)} rows), f32: `profiles/{tag}_ops_table_f32.md`.")
sk = det.get("extras_skipped")
if sk:
    L("")
    L(f"Extras sections skipped for the wall-clock budget in this run: {[s['section'] for s in sk]}.")
