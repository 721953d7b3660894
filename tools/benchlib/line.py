"""The ONE stdout line of a run (< 8 000 bytes, every string <= 120 characters) and the detail file behind it."""
from .common import *   # noqa: F401,F403  (json, os, sys, time, np, torch, dist, ROOT, the roofline constants, progress, event_ms)

LINE_LIMIT = 8000   # bytes of the final stdout line: the driver's record keeps the last 8 KB of output and parses the line from there
STR_LIMIT = 120   # characters of a string the driver's record keeps: longer descriptions live in the detail file


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(full, detail_name="bench_detail.json"):
    """
    The ONE stdout line of a run, built from the full result: what the bench contract names (metric ... config, roofline,
    cpu_baseline) and the flat scalars of `config.results`; the per-leg detail (`legs`, `astar`, `config5_share`, `adi`, the
    `roofline_env` ladder, boundary-call timings, notes) stays in `detail_name`, which the line names.  Always < LINE_LIMIT bytes:
    free text is clipped, and should the scalars ever outgrow the limit the least important groups are dropped (and listed).
    """
    cfg = full["config"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "scaling_measured", "rank_values", "value_per_gpu", "value_spread", "efficiency", "vs_baseline", "data") if k in full}
    if isinstance(line.get("value_spread"), dict):
        line["value_spread"] = {k: v for k, v in line["value_spread"].items() if k != "note"}
    line["dtype"] = _short(full["dtype"], 96)
    line["config"] = {"workload": _short(cfg.get("workload_short") or cfg["workload"], STR_LIMIT), "trees_per_gpu": cfg.get("trees_per_gpu"),
                      "max_states": cfg.get("max_states"), "scramble_depth": cfg.get("scramble_depth"), "parallelism": cfg.get("parallelism"),
                      "timed_region": _short(cfg.get("timed_region_short") or cfg.get("timed_region", ""), STR_LIMIT),
                      "results": dict(cfg.get("results") or {})}
    roof = full.get("roofline") or {}
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "algorithmic_frac", "traffic", "traffic_source", "algorithmic_bytes", "flops_per_launch",
            "algorithmic_flops_per_launch", "ms_per_launch", "ms", "fp32_equivalent_tflops")
    line["roofline"] = {k: (_short(roof.get("kernel_short") or roof[k], STR_LIMIT) if k == "kernel" else
                            _short(roof.get("traffic_source_short") or roof[k] or "", STR_LIMIT) if k == "traffic_source" else roof[k]) for k in keep if k in roof}
    for sub in ("env_multi_rotate_2p24", "astar_dominant_kernel", "adi_dominant_kernel", "adi_env"):
        if sub in roof:
            line["roofline"][sub] = {k: (_short(roof[sub].get("kernel_short") or v, STR_LIMIT) if isinstance(v, str) else v)
                                     for k, v in roof[sub].items() if k in keep}
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {k: (_short(cpu.get("sample_short") or cpu[k], STR_LIMIT) if k == "sample" else cpu[k])
                                for k in ("value", "unit", "cores", "host_cpus", "kind", "sample", "env_ops", "bfs_config1", "adi") if k in cpu}
    line["detail"] = detail_name
    dropped = []
    for victim in (("cpu_baseline", "env_ops"), ("cpu_baseline", "bfs_config1"), ("roofline", "adi_env"), ("roofline", "astar_dominant_kernel")):
        if len(json.dumps(line)) < LINE_LIMIT - 64:
            break
        if victim[1] in line.get(victim[0], {}):
            del line[victim[0]][victim[1]]
            dropped.append(".".join(victim))
    if len(json.dumps(line)) >= LINE_LIMIT - 64:   # last resort: keep the scalars in the order they were added until the line fits
        res, n_cut = line["config"]["results"], 0
        while res and len(json.dumps(line)) >= LINE_LIMIT - 160:
            res.popitem()
            n_cut += 1
        dropped.append(f"config.results: the last {n_cut} scalars")
    if dropped:
        line["dropped_to_detail"] = dropped
    return _clip_strings(line)


def _clip_strings(x):
    if isinstance(x, str):
        return _short(x, STR_LIMIT)
    if isinstance(x, dict):
        return {k: _clip_strings(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_clip_strings(v) for v in x]
    return x


def emit(full, detail_path):
    """Writes the full result to `detail_path` (and to gpurun_out/ when that exists) and prints the compact line, last, on stdout."""
    text = json.dumps(full)
    for path in {detail_path, *([os.path.join(ROOT, "gpurun_out", os.path.basename(detail_path))] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])}:
        try:
            with open(path, "w") as f:
                f.write(text + "\n")
        except OSError as e:   # a read-only checkout must not cost the run its line
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    line = json.dumps(compact_line(full, os.path.basename(detail_path)))
    assert len(line) < LINE_LIMIT and "\n" not in line
    sys.stdout.flush()
    print(line, flush=True)
