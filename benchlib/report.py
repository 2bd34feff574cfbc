"""What bench.py prints: the compact line the driver parses (< 4 KB), the full record beside it, and the SURVEY 8d whole-step / LK rooflines."""
import json
import os
import sys

import numpy as np

from .common import ROOT, KF_EVERY, CULL_FRACTION, HBM_PEAK_GBS, pyramid_bytes


def newest_pmc(S):
    """the newest profiles/r*_pmc_pyramid_batch_s<S>.json (names sort by round + letter), or None"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_pyramid_batch_s{S}.json")))
    return c[-1] if c else None


LK_VISIT_BYTES = lambda w: 3 * (2 * w + 1) ** 2 * 8 + (2 * w + 2) ** 2 * 8 + 12 * 8 + 33     # template + target footprint + 12 corners + point record


def frame_and_lk_rooflines(wl, head, frac3d):
    """SURVEY 8d's whole-step and LK bytes for the headline loop (per stream and key-frame period: KF_EVERY left builds + KF_EVERY temporal
    matches + 1 detect + 1 right build + 1 stereo match), against the measured step / match span.  Level visits per keypoint follow
    map_manager.jl:451-564 + tracker.jl:30-66: a 2-D keypoint = 4 forward + 1 backward visit, a 3-D keypoint with a prior = 2 + 1
    (pyramid_levels_3d = 1); failed 3-D attempts that fall back to the 2-D pass are not counted (a lower bound on the bytes)."""
    S, H, W, levels, params = wl["S"], wl["H"], wl["W"], wl["levels"], wl["params"]
    pb = pyramid_bytes(H, W, levels)
    vb = LK_VISIT_BYTES(params.window_size)
    kpts = head["tracked_kpts_per_frame"]
    visits = frac3d * 3 + (1 - frac3d) * 5
    lk_point = vb * visits
    K = wl["kpts"]
    detect_b = 8 * H * W + 16 * K + 16 * K * CULL_FRACTION
    stereo_b = vb * 5 * K                                     # stereo match: every keypoint as a 2-D keypoint (shift prior, all levels)
    per_period = KF_EVERY * pb + KF_EVERY * kpts * lk_point + detect_b + (pb if wl["stereo"] else 0) + (stereo_b if wl["stereo"] else 0)
    step_bytes = S * per_period
    sec = head["ms_per_step"] * 1e-3
    fr = {"algorithmic_bytes_per_step": int(step_bytes), "achieved": step_bytes / sec / 1e9, "frac": step_bytes / sec / 1e9 / HBM_PEAK_GBS,
          "bound_fps_at_peak": S * KF_EVERY / (step_bytes / (HBM_PEAK_GBS * 1e9)), "visits_per_kpt": round(visits, 2), "frac_3d": round(frac3d, 3)}
    lk = None
    if head.get("lk_match"):
        m = head["lk_match"]
        b = m["points_per_launch"] * lk_point
        lk = {"kernel": "k_kpset_match", "algorithmic_bytes_per_launch": int(b), "avg_launch_us": m["mean_ms"] * 1e3, "points_per_launch": int(m["points_per_launch"]),
              "ns_per_point": m["mean_ms"] * 1e6 / max(m["points_per_launch"], 1), "achieved": b / (m["mean_ms"] * 1e-3) / 1e9,
              "frac": b / (m["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "note": "VALU-issue bound, not HBM (DESIGN 3.3)"}
    return fr, lk


def _r(x, n=4):
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if abs(x) < 1 else round(x, 3)
    return x


def compact_line(out):
    """The ONE stdout line the driver parses: numbers only, < 4 KB (hard limit 8 KB).  Everything else lives in bench_detail.json."""
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    c["config"] = {"workload": "KITTI-05-shaped stereo 370x1226 @1000 kpts, KF every 5th frame (BASELINE configs[1]); step = 1 key-frame period of each stream; "
                               "u8 frames from pinned host memory inside the timed loop; f64 bit-exact",
                   "streams_per_gpu": cfg.get("streams_per_gpu"), "frames_per_step": cfg.get("frames_per_step"), "parallelism": cfg.get("parallelism"),
                   "pyramid_mode": out.get("pyramid_mode", "bit-exact")}
    if out.get("per_rank_values"):
        c["per_rank_values"] = [_r(v) for v in out["per_rank_values"]]
    rf = out.get("roofline")
    if rf:
        c["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_isolated", "algorithmic_bytes_per_launch", "avg_launch_us",
                                                     "isolated_launch_us", "traffic", "traffic_over_algorithmic")}
        c["roofline"]["stage"] = f"LK pyramid update of {cfg.get('streams_per_gpu')} images, one graph launch"
        if rf.get("traffic_source"):
            c["roofline"]["traffic_source"] = rf["traffic_source"].split(" ")[0]
        for k in ("frame", "lk"):
            if rf.get(k):
                c["roofline"][k] = {a: _r(b) for a, b in rf[k].items() if a not in ("note", "kernel")}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:120]}
        if out.get("value"):
            c["cpu_baseline"]["gpu_over_cpu"] = _r(out["value"] / cb["value"])      # 128 lock-stepped streams vs ONE CPU stream
        live = (out.get("single_stream") or {}).get("by_builds_in_flight", {}).get("1")
        if live:
            c["cpu_baseline"]["gpu_over_cpu_one_live_stream"] = _r(live / cb["value"])      # like for like: one stream, next frame only
    ba = out.get("ba")
    if ba:
        c["ba"] = {"ms_per_iter": _r(ba.get("ms_per_iter")), "window_kf": 50, "observations": ba.get("observations"),
                   "windows_ms_per_iter": {k: _r(v["ms_per_iter"]) for k, v in ba.get("windows", {}).items()},
                   "roofline_frac_P50": _r(ba.get("windows", {}).get("P50", {}).get("roofline", {}).get("frac")),
                   "cpu_ms_per_iter_schur": _r(ba.get("cpu_ms_per_iter_schur")), "cpu_ms_per_iter_lm_lsmr": _r(ba.get("cpu_ms_per_iter_reference_style_lm_lsmr"))}
    if ba and ba.get("batch"):
        c["ba"]["batch"] = {k: {"windows": v["windows"], "wall_ms": _r(v["wall_ms_per_call"]), "device_ms": _r(v["device_ms_per_call"]), "windows_per_s": _r(v["windows_per_s"]),
                                **({"windows_per_s_two_in_flight": _r(v["windows_per_s_two_calls_in_flight"])} if isinstance(v.get("windows_per_s_two_calls_in_flight"), float) else {})}
                            for k, v in ba["batch"].items()}
    fb = out.get("frontend_with_ba")
    if fb:
        c["frontend_with_ba"] = ({"value": _r(fb["value"]), "fraction_of_headline": _r(fb.get("fraction_of_headline")), "ba_call_ms": _r(fb["local_ba"]["mean_call_ms"]),
                                  "windows_per_call": fb["local_ba"]["windows_per_call"]} if "value" in fb else {"error": fb.get("error", "")[:120]})
    bs = out.get("ba_sharded")
    if bs:
        c["ba_sharded"] = {k: _r(bs.get(k)) for k in ("world_size", "window_kf", "ms_per_iter_wall", "worth_sharding", "rccl_ranks_seen", "error") if bs.get(k) is not None}
        if bs.get("predicted_crossover_keyframes"):
            c["ba_sharded"]["predicted_crossover_keyframes"] = bs["predicted_crossover_keyframes"]      # the worth_sharding model's window size from which N GPUs pay, per N
    ss = out.get("single_stream")
    if ss and "by_builds_in_flight" in ss:
        c["single_stream"] = {"live": _r(ss["by_builds_in_flight"].get("1")), "lookahead": _r(ss.get("value")), "unit": "frames/sec"}
        if isinstance(ss.get("live_step"), dict) and "value" in ss["live_step"]:
            # "next frame only" has two implementations: one C call per frame on device-resident lists (slam_frontend_step) and the six seams called from
            # Python on host lists; `live` is the better of the two, both are on the line
            c["single_stream"]["live_one_call_per_frame"] = _r(ss["live_step"]["value"])
            c["single_stream"]["live_python_protocol"] = _r(ss["by_builds_in_flight"].get("1_python_protocol"))
            c["single_stream"]["live"] = _r(max(ss["live_step"]["value"], ss["by_builds_in_flight"].get("1_python_protocol") or 0.0))
        if ss.get("live_graph") is not None:
            c["single_stream"]["live_graph"] = _r(ss["live_graph"])
    elif ss:
        c["single_stream"] = {"error": str(ss.get("error"))[:120]}
    tm = out.get("tolerance_mode")
    if tm:
        c["tolerance_mode"] = {k: _r(v) for k, v in tm.items() if isinstance(v, (int, float, bool))}
        if isinstance(tm.get("single_stream"), dict):
            c["tolerance_mode"]["single_stream"] = _r(tm["single_stream"].get("value"))
        if isinstance(tm.get("single_stream_live"), dict):
            c["tolerance_mode"]["single_stream_live"] = _r(tm["single_stream_live"].get("value"))
        if isinstance(tm.get("replayed_frames"), dict):
            rf_ = tm["replayed_frames"]
            c["tolerance_mode"]["replayed_frames"] = ({"ok": rf_["ok"], "max_dpx": _r(rf_["max_abs_position_diff_px"]), "fate_flips": _r(rf_["fate_flip_fraction"])} if "ok" in rf_
                                                      else {"error": rf_.get("error", "")[:100]})
        if isinstance(tm.get("batch"), dict):
            b = tm["batch"]
            c["tolerance_mode"].update({"value": _r(b.get("value")), "ms_per_step": _r(b.get("ms_per_step")), "planes_rel_tol": 1e-11})
            c["tolerance_mode"]["roofline"] = {k: _r(b["roofline"].get(k)) for k in ("frac", "frac_isolated", "avg_launch_us", "isolated_launch_us", "traffic", "traffic_over_algorithmic")}
    if out.get("configs"):
        c["configs"] = {k: (_r(v.get("value")) if "value" in v else "error") for k, v in out["configs"].items()}
        wb = {k: ({"value": _r(v["with_ba"]["value"]), "fraction_of_front_end_only": _r(v["with_ba"]["fraction_of_front_end_only"]), "ba_call_ms": _r(v["with_ba"]["ba_call_ms"]),
                   "ba_window": v["with_ba"]["ba_window"], "ok": v["with_ba"]["all_windows_ok"]} if "value" in v["with_ba"] else {"error": v["with_ba"].get("error", "")[:100]})
              for k, v in out["configs"].items() if isinstance(v.get("with_ba"), dict)}
        if wb:
            c["with_ba"] = wb                                      # each BASELINE config as named: the front-end loop with its 20 / 50 / 100-KF local BA per stream and key-frame
        if any("tolerance_value" in v for v in out["configs"].values()):
            c["configs_tolerance_mode"] = {k: _r(v.get("tolerance_value")) for k, v in out["configs"].items() if "tolerance_value" in v}
    if out.get("pose", {}).get("frontend_with_pose"):
        c["frontend_with_pose"] = _r(out["pose"]["frontend_with_pose"]["value"])
    # the loops that recover poses check them against the scene (translation within pose_tol_m, every compute_pose! accepted)
    pk = {}
    if out.get("pose", {}).get("frontend_with_pose"):
        f = out["pose"]["frontend_with_pose"]
        pk["frontend_with_pose"] = {"ok": f.get("pose_ok"), "max_translation_error_m": _r(f.get("max_translation_error_m"))}
    for k, v in (out.get("configs") or {}).items():
        if isinstance(v.get("pose"), dict):
            pk[k] = {"ok": v["pose"].get("pose_ok"), "max_translation_error_m": _r(v["pose"].get("max_translation_error_m"))}
    if pk:
        c["pose_ok"] = all(bool(v["ok"]) for v in pk.values())
        c["pose_checks"] = pk
    pv = out.get("parity_vs_oracle")
    if pv:
        c["parity_vs_oracle"] = {"ok": pv["ok"] and not out.get("parity_failures")}
    if out.get("parity_failures"):
        c["parity_failures"] = len(out["parity_failures"])
    if out.get("leg_error"):
        c["leg_error"] = {"after_leg": out["leg_error"].get("after_leg"), "error": str(out["leg_error"].get("error"))[:160]}
    c["detail"] = "bench_detail.json"
    line = json.dumps(c, separators=(",", ":"))
    if len(line) > 8000:                                          # never lose the line to its own size: drop the optional objects, largest first
        for k in ("configs", "ba_sharded", "tolerance_mode", "single_stream"):
            c.pop(k, None)
        line = json.dumps(c, separators=(",", ":"))
    assert len(line) <= 8000, len(line)
    return line


def write_detail(out):
    """the full record (notes, sweeps, per-window objects): next to bench.py, under gpurun_out/ when that exists, and on stderr"""
    txt = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(txt + "\n")
            except OSError:
                pass
    print(txt, file=sys.stderr, flush=True)


