"""Print the DESIGN.md results section from a bench JSON line (profiles/r01_bench_<cfg>.json)."""
import json
import sys

d = json.load(open(sys.argv[1]))
print(f"`python bench.py` ({d['config']['workload']}, 1 GPU, {d['steps']} timed LM iterations):")
print(f"**{d['value']} LM iterations/s** ({d['ms_per_step']} ms per iteration); one complete solve: "
      f"{d['solve']['iterations']} iterations, {d['solve']['termination']}, RMSE {d['solve']['rmse_px']} px, "
      f"{d['solve']['solve_seconds']} s solve + {d['solve']['setup_seconds']} s host set-up.")
cb = d.get("cpu_baseline")
if cb:
    print(f"CPU baseline on the same box ({cb['cores']} host threads): {cb['value']} {cb['unit']} ({cb['sample']}).")
js = d["jacobian_sweep"]
print(f"\nJacobian sweep: {js['obs_per_sec']:.3g} observations/s, {js['achieved']} GB/s of algorithmic bytes = "
      f"{100 * js['frac']:.1f} % of the 8 TB/s HBM peak" + (f"; PMC traffic {js['traffic']:.3g} B per launch." if js.get("traffic") else "."))
rs = d["reduced_system"]
print(f"\nReduced system: n = {rs['n']} ({rs['matrix_dim']} matrix columns), {rs['nd_parts']} concurrent fronts, "
      f"{rs['chain_steps']} dependent panel steps, {rs['envelope_tiles']} of {rs['dense_tiles']} tiles, "
      f"{rs['factor_gflop_envelope']} GFLOP executed ({rs['factor_gflop_dense_equivalent']} dense-equivalent); "
      f"{rs['schur_clusters']} point clusters cover {rs['clustered_points']} points and emit {rs['cluster_partials']} block partials.")
print("\n| timer (HIP events in the timed region) | launches | avg ms | share | rate (fraction of peak) |")
print("|---|---:|---:|---:|---|")
for k in d["kernels"]:
    if k["share"] < 0.004:
        continue
    rate = f"{k['achieved']} {k['unit']} ({100 * k['frac']:.1f} %)" if k.get("bound") else "—"
    print(f"| `{k['kernel']}` | {k['launches']} | {k['avg_ms']:.4f} | {100 * k['share']:.1f} % | {rate} |")
