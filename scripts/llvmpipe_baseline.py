#!/usr/bin/env python3
"""The reference's own CPU path timed on the BASELINE workloads (development container only).

north_star: "... next to the reference timed on the host CPU (OpenGL llvmpipe / Mesa software path, core count
stated)".  The reference's C++ cannot be built here (ROS, tf, urdfdom, Assimp, OpenCV, GLEW, freeglut are absent),
but its GPU programs can run: oracle/ref_gl/llvmpipe_oracle.c replays the GL call sequence of
src/urdf_filter.cpp:332-353, :503-744 with the reference's verbatim urdf_filter.vert/.frag (read from /root/reference
at run time) on Mesa llvmpipe.  Per frame this script times exactly what RealtimeURDFFilter::filter() does per frame
(src/urdf_filter.cpp:211-244): depth upload into the texture buffer, render of every link from STATIC vertex /
index buffers (created once, like the Renderable constructors do), and the two glGetTexImage read-backs.

llvmpipe's thread count is fixed when the GL context is created (LP_NUM_THREADS), so every (workload, threads) pair
runs in its own process:   python scripts/llvmpipe_baseline.py > profiles/llvmpipe_baseline.json
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "C1": "640x480, urdf/example.urdf.xml (two boxes incl. quirk Q1), one stream",
    "C2/C3": "640x480, synthetic PR2-like URDF, 250,388 triangles (the bench.py model), one frame = one stream of the batch",
    "C2-small": "640x480, synthetic PR2-like URDF, ~20 k triangles (collision-mesh sized)",
    "C4": "1280x720, synthetic PR2-like URDF (250 k triangles) + the two wall URDFs",
}


def build_workload(name, frames):
    from bench_support import workloads as WL
    if name == "C1":
        wl = WL.example_workload(640, 480)
        return wl, 1
    if name == "C2/C3":
        return WL.pr2_workload(frames, 640, 480, total_triangles=250000), frames
    if name == "C2-small":
        return WL.pr2_workload(frames, 640, 480, total_triangles=20000), frames
    if name == "C4":
        return WL.pr2_workload(frames, 1280, 720, total_triangles=250000, walls=True), frames
    raise SystemExit("unknown workload " + name)


def child(name, frames, seconds, shaders="reference"):
    import numpy as np
    from oracle.ref_gl import harness as HN
    from oracle import bindings as O
    wl, n_states = build_workload(name, frames)
    h = HN.Harness(wl.width, wl.height, shaders)
    L = h.L
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    # static geometry once: one VBO/IBO per draw call (src/renderable.cpp:167-169, :343-349)
    L.rgo_mesh_clear()
    keep = []
    links = []          # (model index, link index, [(mesh id, pre_op, op)])
    for mi, model in enumerate(wl.models):
        for li, draws in enumerate(model):
            ids = []
            for d in draws:
                v = np.ascontiguousarray(d.verts, np.float32).reshape(-1, 3)
                t = np.ascontiguousarray(d.tris, np.uint32).reshape(-1)
                keep += [v, t]
                ids.append((L.rgo_mesh_create(p(v), len(v), 3, p(t), t.size), d.pre_op, [float(x) for x in d.op]))
            links.append((mi, li, ids))
    depth = [np.ascontiguousarray(wl.depth(s), np.float32) for s in range(n_states)]
    out = np.zeros((wl.height, wl.width), np.float32)
    mask = np.zeros((wl.height, wl.width), np.uint8)

    def frame(s):
        P = np.ascontiguousarray(wl.projection[s], np.float64)
        oi = np.ascontiguousarray(wl.offset_inv[s], np.float64)
        ct = np.ascontiguousarray(wl.cam_tf[s], np.float64)
        L.rgo_begin_frame(p(depth[s]), p(P), p(oi), p(ct), ctypes.c_float(wl.near), ctypes.c_float(wl.far), ctypes.c_float(wl.max_diff), ctypes.c_float(wl.replace_value))
        for mi, li, ids in links:
            tf = np.ascontiguousarray(wl.link_tf[mi][s, li], np.float64)
            L.rgo_push_link(p(tf))
            for mid, pre, op in ids:
                if pre == 1:
                    L.rgo_scale(*op)
                elif pre == 2:
                    L.rgo_translate(*op)
                L.rgo_mesh_draw(mid, HN.GL_TRIANGLES)
            L.rgo_pop_link()
        L.rgo_end_frame(p(out), p(mask))

    frame(0)            # shader JIT, first-touch
    # the harness result equals the oracle's on this workload (what "the reference's result" means in this project)
    om, ok = O.filter_frame(depth[0], wl.projection[0], wl.oracle_draws(0), wl.offset_inv[0], wl.cam_tf[0], max_diff=wl.max_diff, replace_value=wl.replace_value)
    same = bool(np.array_equal(ok, mask) and np.array_equal(om.view(np.uint32), out.view(np.uint32)))
    t0 = time.perf_counter()
    n = 0
    while n < 3 or time.perf_counter() - t0 < seconds:
        frame(n % n_states)
        n += 1
    el = time.perf_counter() - t0
    print(json.dumps({"workload": name, "description": WORKLOADS[name], "width": wl.width, "height": wl.height, "triangles": wl.n_triangles(),
                      "threads": int(os.environ.get("LP_NUM_THREADS", "0")), "renderer": h.renderer(), "shaders": shaders,
                      "frames": n, "seconds": el, "frames_per_s": n / el, "ms_per_frame": el / n * 1e3, "equals_oracle": same}))


def run_children(name, frames, seconds, shaders, threads, procs):
    """`procs` concurrent processes (one GL context each, like one reference node per camera) with `threads` llvmpipe
    threads each; returns the list of their result dicts."""
    env = dict(os.environ, LP_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), "--child", name, "--frames", str(frames), "--seconds", str(seconds), "--shaders", shaders]
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT) for _ in range(procs)]
    out = []
    for pr in ps:
        so, se = pr.communicate(timeout=200)
        line = [l for l in so.splitlines() if l.startswith("{")]
        out.append(json.loads(line[-1]) if (pr.returncode == 0 and line) else {"error": (se or so)[-300:]})
    return out


def bench_leg(args):
    """The llvmpipe figures bench.py embeds in cpu_baseline: the bench workload (C2/C3 model), one stream per frame."""
    cores = len(os.sched_getaffinity(0))
    lp_max = min(16, cores)                    # llvmpipe caps its rasteriser threads (LP_MAX_THREADS = 16 in Mesa 23)
    procs = max(1, cores // lp_max)
    res = {"cpu": cpu_model(), "cores_available": cores, "shaders": args.shaders, "seconds_per_configuration": args.seconds}
    legs = [("one_process_1_thread", 1, 1), ("one_process_%d_threads" % lp_max, lp_max, 1)]
    if procs > 1:
        legs.append(("%d_processes_x_%d_threads" % (procs, lp_max), lp_max, procs))
    res["configurations"] = [l[0] for l in legs]
    for label, threads, np_ in legs:
        rs = run_children("C2/C3", args.frames, args.seconds, args.shaders, threads, np_)
        good = [r for r in rs if "frames_per_s" in r]
        res[label] = {"frames_per_s": sum(r["frames_per_s"] for r in good), "processes": np_, "threads_per_process": threads,
                      "processes_failed": len(rs) - len(good), "equals_oracle": all(r.get("equals_oracle") for r in good) if good else None,
                      "renderer": good[0]["renderer"] if good else None, "error": None if good else rs[0].get("error")}
    print(json.dumps(res))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--frames", type=int, default=8, help="distinct joint states / sensor frames cycled through")
    ap.add_argument("--seconds", type=float, default=6.0, help="timed seconds per (workload, thread count)")
    ap.add_argument("--threads", type=int, nargs="*", default=None)
    ap.add_argument("--shaders", choices=["reference", "standin"], default="reference",
                    help="reference: the reference's own GLSL from /root/reference (development container); standin: oracle/ref_gl/standin_shaders (bit-identical re-statement, for machines without /root/reference)")
    ap.add_argument("--bench-leg", action="store_true", help="bench.py's cpu_baseline leg: the bench workload only, 1 thread / one process at llvmpipe's thread limit / the whole box")
    args = ap.parse_args()
    if args.child:
        child(args.child, args.frames, args.seconds, args.shaders)
        return
    if args.bench_leg:
        bench_leg(args)
        return
    cores = len(os.sched_getaffinity(0))
    threads = args.threads or sorted({1, min(8, cores), cores})
    results = []
    for name in WORKLOADS:
        for t in threads:
            env = dict(os.environ, LP_NUM_THREADS=str(t))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--frames", str(args.frames), "--seconds", str(args.seconds)],
                               capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                results.append({"workload": name, "threads": t, "error": (r.stderr or r.stdout)[-400:]})
                continue
            results.append(json.loads(line[-1]))
            print("%-9s threads=%-3d %8.2f frames/s  %8.2f ms/frame  equals_oracle=%s" % (name, t, results[-1]["frames_per_s"], results[-1]["ms_per_frame"], results[-1]["equals_oracle"]), file=sys.stderr)
    best = max((r for r in results if r.get("workload") == "C2/C3" and "frames_per_s" in r), key=lambda r: r["frames_per_s"], default=None)
    one = next((r for r in results if r.get("workload") == "C2/C3" and r.get("threads") == 1 and "frames_per_s" in r), None)
    out = {"what": "the reference's GLSL (include/shaders/urdf_filter.{vert,frag}, read from /root/reference at run time) on Mesa llvmpipe with the reference's per-frame GL call sequence: depth upload + render from static VBOs + two glGetTexImage read-backs (src/urdf_filter.cpp:211-244)",
           "where": "development container (no GPU); the GPU box has no /root/reference", "cpu": cpu_model(), "cores_available": cores,
           "thread_counts": threads, "results": results,
           "bench_workload": {"workload": "C2/C3 (the bench.py model, one stream per frame)",
                              "frames_per_s_1_thread": one["frames_per_s"] if one else None,
                              "frames_per_s_best": best["frames_per_s"] if best else None, "threads_best": best["threads"] if best else None,
                              "cpu": cpu_model(), "cores_available": cores, "renderer": best["renderer"] if best else None}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
