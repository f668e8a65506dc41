"""Model (no GPU needed): what a BRICK-MAJOR ray-march would cost at C3 ("model it first; build only if it says >= 15 % of 0.94 ms").
One workgroup per (brick, screen footprint), placed on the XCD that owns the brick's slot, so that every brick is fetched into ONE L2 instead of
up to eight; rays are cut into per-brick segments whose partial results are composited afterwards in the reference's global order.
Inputs: a bench line (profiles/r06_bench_lines/bench_C3_r8.json or any C3 line), the phase profile of round 3 (sample loops 63.6 % of the wave time,
cell walk 28.1 %: profiles/r03_raymarch_phase_profile_C3_r8.txt) and the counter traffic of the ray-march (profiles/traffic_C3_r8.json).
usage: raymarch_brick_major_model.py <bench_line.json> [traffic.json]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c, r = d["config"], d["roofline_all"]["raymarch"]
t_ms, alg = r["avg_ms"], r["bytes_per_launch"]
traffic = json.load(open(sys.argv[2]))["k_raymarch"]["traffic_bytes"] if len(sys.argv) > 2 else 2.356e9
s_exec, s_formula = c["samples_executed"], c["samples_formula"]
steps = 64
per_unit = steps / 1.73205            # lattice samples per metavoxel unit along a ray (RM.shader: mvStep = 1.73205 / steps)
chord = 4.0 * 1.0 / 6.0               # mean chord of a unit cube over uniformly random lines = 4 V / S
per_visit = per_unit * chord
v_formula, v_exec = s_formula / per_visit, s_exec / per_visit
print(f"C3 ray-march today: {t_ms:.3f} ms, {s_exec / 1e6:.0f} M samples executed of {s_formula / 1e6:.0f} M formula samples (saturation early-out), "
      f"algorithmic {alg / 1e9:.3f} GB, counter traffic {traffic / 1e9:.3f} GB = {traffic / alg:.2f}x (re-fetch by the eight L2s: {(traffic - alg) / 1e9:.2f} GB)")
print(f"HBM rate today {traffic / t_ms / 1e9:.2f} TB/s of ~6.3 achievable: the launch is not memory-paced (VALU issue 0.57, lanes 0.69: latency-bound visit chains)")
print(f"(pixel, metavoxel) visits: ~{per_visit:.1f} samples each -> {v_formula / 1e6:.1f} M visits without an early-out, {v_exec / 1e6:.1f} M with today's")
part = 16
extra = v_formula * part * 2
print(f"brick-major: every visit leaves a partial (premultiplied rgb, transmittance) = {part} B written and read back by an ordered resolve pass: "
      f"{extra / 1e9:.2f} GB of new traffic against {(traffic - alg) / 1e9:.2f} GB of re-fetch removed -> {(traffic - alg - extra) / 1e9:.2f} GB saved net ({(traffic - alg - extra) / traffic:.0%} of today's traffic)")
loops, rest = 0.636, 1.0 - 0.636
no_early = rest + loops * s_formula / s_exec
print(f"a brick's workgroup cannot know what the bricks in front of it left of a ray: every formula sample is executed ({s_formula / s_exec:.2f}x); "
      f"sample loops are {loops:.1%} of the wave time -> x{no_early:.2f}")
resolve = (v_formula * part + 16 * 1920 * 1080) / 4.0e12 * 1e3
pred = t_ms * no_early + resolve
print(f"resolve pass (ordered blend of <= ~30 partials per pixel, {v_formula * part / 1e9:.2f} GB at ~4 TB/s): +{resolve:.2f} ms")
print(f"PREDICTED {pred:.2f} ms vs {t_ms:.3f} ms today ({(pred / t_ms - 1) * 100:+.0f} %): NOT built.  (Keeping the early-out needs front-to-back phases with a "
      f"saturation map between them -- the fan-out's hand-off, measured in round 3/4 to save samples but not time -- and each phase boundary is a device-wide barrier.)")
