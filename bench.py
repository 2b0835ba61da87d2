#!/usr/bin/env python3
"""Headline benchmark: whole-node Msamples/s of the path-tracing hot path (BASELINE.json).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one full pass of the hot path over the workload: BASELINE config B, the reference's default
scene (scene.rs:43-66) at 1920x1080 with 1024 samples per pixel PER GPU (weak scaling: at N GPUs the
film is rendered at 1024*N spp, rank r takes sample indices [1024 r, 1024 (r+1)), and the partial films
are summed with one RCCL all-reduce inside the timed region).  Scene, camera and film live in HBM before
the timed region starts; nothing is copied to the host inside it.

Msamples/s = W*H*spp / 1e6 / seconds, the definition main.rs:109-114 prints.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flop model (DESIGN.md "Roofline"): FMA = 2, sqrt/div/sin/cos = 1.
F_CAM, F_ISECT, F_FRAME, F_SHADE = 120.0, 27.0, 44.0, 400.0
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (vector)"
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md, HBM3E bandwidth
# roofline.traffic is never typed in: it is read from profiles/traffic.json, which tools/summarize_profile.py writes from
# the PMC passes of this same command (TCC_EA0_WRREQ/RDREQ, corrected as MI355X_MICROARCH.md prescribes), keyed by the
# workload; a workload without a committed PMC profile reports null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")


def measured_profile(key):
    """The committed PMC profile of a workload (profiles/traffic.json, written by tools/summarize_profile.py), or {}."""
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f).get(key) or {}
    except (OSError, ValueError):
        return {}


def measured_traffic(key):
    e = measured_profile(key)
    try:
        return (float(e["bytes_per_launch"]), e["source"]) if e else (None, None)
    except KeyError:
        return None, None


def workload_key(scene, width, height, spp, opts=()):
    return "%s_%dx%dx%d%s" % (scene, width, height, spp, "".join("_%s=%d" % kv for kv in sorted(dict(opts).items())))


def make_scene(iris_amd, name, env_size):
    scene = {"s1": iris_amd.Scene.dummy, "s2": iris_amd.Scene.many_spheres, "s3": iris_amd.Scene.glossy_dielectric}[name[:2]]()
    if name.endswith("env"):  # procedural HDR sky through the RGB -> spectrum table (a data fixture; tools/rgb2spec_opt.c makes it)
        table = iris_amd.UpsampleTable(os.path.join(ROOT, "tests", "golden", "srgb_res16.coeff"))
        scene.env_map = iris_amd.EnvMap.sky(table, env_size, env_size)
    return scene


def roofline_of(iris_amd, scene, key, width, height, steps, st, kernel_ms, flops):
    """The `roofline` object of one workload from the exact device counters and the HIP-event kernel time of `steps` steps."""
    n_launches = steps * max(1, st.chunks)
    secs = kernel_ms * 1e-3
    # flops the reference's algorithm executes for this film, and the part of them the kernel executes: the samples of
    # pixels that the sky mask proves black are never traced (their camera sample and primary scan are not run)
    flops_skipped = steps * (F_CAM + F_ISECT * len(scene.primitives)) * st.sky_samples
    executed = (flops - flops_skipped) / secs / 1e12 if secs > 0 else 0.0
    reference_work = flops / secs / 1e12 if secs > 0 else 0.0
    prof = measured_profile(key)
    traffic, traffic_source = measured_traffic(key)
    # algorithmic bytes = the film (12 B per pixel) + the scene read once (16 B per sphere)
    algorithmic_bytes = 12.0 * width * height + 16.0 * len(scene.primitives)
    return {"bound": "valu", "achieved": round(executed, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / PEAK_FP32_VALU_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes": algorithmic_bytes,
            "traffic_over_algorithmic": round(traffic / algorithmic_bytes, 1) if traffic else None,
            "kernel": "%s<HwssNaive>" % iris_amd.ffi.KERNEL_NAMES[st.kernel], "kernel_ms_per_launch": round(kernel_ms / n_launches, 3),
            "flop_model": "120*samples + 27*sphere_tests + 444*bounces over exact device counters (= the oracle's), minus "
                          "(120 + 27*n_prims)*sky_samples for the camera samples the sky mask proves black and never traces",
            "achieved_reference_work": round(reference_work, 3),
            "frac_reference_work": round(reference_work / PEAK_FP32_VALU_TFLOPS, 4),
            # read from the committed rocprofv3 PMC profile of this workload (profiles/, like `traffic`); null without one
            "valu_busy": prof.get("valu_busy"), "lane_utilisation": prof.get("lane_utilisation"),
            "wave_cycles_waiting": prof.get("wave_cycles_waiting"), "counters_source": prof.get("counters_source")}


def other_config(iris_amd, torch, name, label, width, height, spp, steps, env_size, device):
    """One more BASELINE configuration on this GPU, measured like the headline: inputs resident, `steps` timed passes."""
    scene = make_scene(iris_amd, name, env_size)
    max_depth = 32 if name.startswith("s3") else iris_amd.scene.MAX_DEPTH
    render = iris_amd.Render(width, height, spp, scene, max_depth=max_depth)
    film = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    with iris_amd.HipWavefront(device=device) as gpu:
        gpu.upload(scene)
        gpu.render_film_device(render, film.data_ptr())  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kernel_ms, flops, st = 0.0, 0.0, None
        for _ in range(steps):
            gpu.render_film_device(render, film.data_ptr())
            st = gpu.stats()
            kernel_ms += st.kernel_ms
            flops += algorithmic_flops(st)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    return {"workload": label, "metric": "Msamples/s", "value": round(width * height * spp * steps / 1e6 / elapsed, 2),
            "value_traced_only": round((st.samples - st.sky_samples) * steps / 1e6 / elapsed, 2),
            "ms_per_step": round(elapsed * 1e3 / steps, 3), "steps": steps, "kernel": "%s<HwssNaive>" % iris_amd.ffi.KERNEL_NAMES[st.kernel],
            "kernel_ms_per_launch": round(kernel_ms / (steps * max(1, st.chunks)), 3),
            "roofline": roofline_of(iris_amd, scene, workload_key(name, width, height, spp), width, height, steps, st, kernel_ms, flops),
            "rays_per_s": round((st.rays - st.sky_samples) / (kernel_ms / steps * 1e-3), 1) if kernel_ms else None,
            "counters_per_step": {"samples": st.samples, "bounces": st.bounces, "rays": st.rays, "sphere_tests": st.sphere_tests,
                                  "bvh_depth": st.bvh_depth, "sky_samples": st.sky_samples}}


def algorithmic_flops(st):
    return F_CAM * st.samples + F_ISECT * st.sphere_tests + (F_FRAME + F_SHADE) * st.bounces


def host_cpu():
    model, cores = "unknown", os.cpu_count() or 1
    try:
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("model name"):
                    model = l.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, cores


def oracle_build_flags():
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            for l in f:
                if l.startswith("CFLAGS"):
                    return l.split("=", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_threads):
    """The oracle (a C port of the reference's CPU path, oracle/iris_oracle.c) timed on this host's cores
    on BASELINE config A: 512x512 at 64 spp, default scene, HwssNaive."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_ffi as O
    from iris_amd import Scene
    O.set_trig_mode(O.TRIG_LIBM)
    w = h = 512
    spp = 64
    sc = O.OracleScene(Scene.dummy().flatten())
    cam = O.camera_new([0, 0, 0], np.float32(1.0))
    reps = 3  # about 25 core-seconds in all
    secs = sum(sc.render(cam, O.make_params(w, h, spp), nthreads=n_threads)[2] for _ in range(reps))
    model, total = host_cpu()
    return {"value": round(reps * w * h * spp / 1e6 / secs, 3), "unit": "Msamples/s", "cores": n_threads, "kind": "port",
            "host_cpu": model, "host_logical_cpus": total, "threads": n_threads, "build": "gcc " + oracle_build_flags(),
            "sample": "BASELINE config A three times over: default scene 512x512 at 64 spp (3 x 16.8 Msamples), HwssNaive, "
                      "oracle C port, %d threads popping 64x64 tiles, %.2f s in all" % (n_threads, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU")
    ap.add_argument("--scene", default="s1", choices=["s1", "s2", "s3", "s1env", "s3env"],
                    help="s1 default scene, s2 10k spheres, s3 glossy/dielectric; ...env: the same under the HDR sky (config E = s3env)")
    ap.add_argument("--env-size", type=int, default=1024, help="environment map is env_size x env_size texels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-c", action="store_true", help="skip the extra strong-scaling leg (BASELINE config C: 3840x2160 at 4096 spp in all)")
    ap.add_argument("--no-relaxed", action="store_true", help="skip the extra leg on the opt-in relaxed-arithmetic build (reported beside `value`, never as it)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra single-GPU legs on BASELINE configs D and E, S3 and the default scene under the sky")
    ap.add_argument("--multi-kernel", action="store_true", help="launch the two stages as separate kernels")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="iris_ctx_set_option tuning knob (repeatable)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import iris_amd
    from iris_amd.distributed import sample_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, args.gpus))
    # Rehearsal on a one-GPU box only: IRIS_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and the film exchange on
    # gloo (RCCL refuses two ranks on one GPU).  The driver's multi-GPU runs never set it.
    share = os.environ.get("IRIS_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        # communicator set-up and the first collective's one-off costs stay out of the timed steps even with --warmup 0
        dist.all_reduce(torch.zeros(1, device="cuda"))
        torch.cuda.synchronize()
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    scene = make_scene(iris_amd, args.scene, args.env_size)
    max_depth = 32 if args.scene.startswith("s3") else iris_amd.scene.MAX_DEPTH
    spp_total = args.spp * world
    render = iris_amd.Render(args.width, args.height, spp_total, scene, max_depth=max_depth)
    opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.opt}
    if args.multi_kernel:
        opts["multi_kernel"] = 1
    gpu = iris_amd.HipWavefront(device=local_rank, **opts)
    gpu.upload(scene)
    film = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device="cuda")
    begin, count = sample_range(spp_total, rank, world)

    def step():
        gpu.render_film_device(render, film.data_ptr(), spp_begin=begin, spp_count=count)
        if world > 1:
            dist.all_reduce(film, op=dist.ReduceOp.SUM)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, flops, st = 0.0, 0.0, None
    resolve_ms, resolve_bytes = 0.0, 0.0
    for _ in range(args.steps):
        step()
        st = gpu.stats()
        kernel_ms += st.kernel_ms
        resolve_ms += st.resolve_ms
        resolve_bytes += 12.0 * (st.samples - st.sky_samples) + 13.0 * args.width * args.height
        flops += algorithmic_flops(st)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- second leg, same ranks: BASELINE config C, STRONG scaling -- the default scene at 3840x2160 with 4096 spp IN ALL,
    # rank r renders sample indices [4096 r / N, 4096 (r+1) / N) of every pixel, one all-reduce of the 99.5 MB film.  Reported
    # as an extra key so that one multi-GPU run of this script yields both the weak (config B per GPU) and the strong curve.
    config_c = None
    if args.scene == "s1" and not args.no_config_c and not opts:
        cw, ch, cspp = 3840, 2160, 4096
        crender = iris_amd.Render(cw, ch, cspp, scene, max_depth=max_depth)
        cfilm = torch.zeros((ch, cw, 3), dtype=torch.float32, device="cuda")
        cbegin, ccount = sample_range(cspp, rank, world)

        def cstep():
            gpu.render_film_device(crender, cfilm.data_ptr(), spp_begin=cbegin, spp_count=ccount)
            if world > 1:
                dist.all_reduce(cfilm, op=dist.ReduceOp.SUM)

        csteps = max(1, min(args.steps, 2))
        if args.warmup:
            cstep()
        fence()
        c0 = time.perf_counter()
        for _ in range(csteps):
            cstep()
        fence()
        celapsed = time.perf_counter() - c0
        if world > 1:
            t = torch.tensor([celapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            celapsed = float(t.item())
        config_c = {"workload": "BASELINE config C: default scene 3840x2160, 4096 spp in all, %d spp per rank, spp-split + one all-reduce of the XYZ film" % ccount,
                    "metric": "Msamples/s", "value": round(cw * ch * cspp * csteps / 1e6 / celapsed, 2), "scaling": "strong",
                    "ms_per_step": round(celapsed * 1e3 / csteps, 3), "steps": csteps, "n_gpus": world,
                    "ranks": dist.get_world_size() if world > 1 else 1}
        del cfilm

    # ---- third leg, one GPU only: the same workload on the opt-in relaxed-arithmetic build (libiris_hip_relaxed.so: 1-ulp
    # reciprocals/roots and contraction in the radiometric code; camera samples and sphere tests exact).  Its film meets the
    # statistical image gate (tests/test_gpu_relaxed.py), not the reference's bits, so it is reported BESIDE the headline value.
    relaxed = None
    if world == 1 and not args.no_relaxed and os.path.exists(iris_amd.ffi.LIB_RELAXED_PATH):
        rgpu = iris_amd.HipWavefront(device=local_rank, relaxed=True, **opts)
        rgpu.upload(scene)
        rsteps = max(1, min(args.steps, 2))
        if args.warmup:
            rgpu.render_film_device(render, film.data_ptr(), spp_begin=begin, spp_count=count)
        fence()
        r0 = time.perf_counter()
        r_kernel_ms, r_flops, rst = 0.0, 0.0, None
        for _ in range(rsteps):
            rgpu.render_film_device(render, film.data_ptr(), spp_begin=begin, spp_count=count)
            rst = rgpu.stats()
            r_kernel_ms += rst.kernel_ms
            r_flops += algorithmic_flops(rst) - (F_CAM + F_ISECT * len(scene.primitives)) * rst.sky_samples
        fence()
        relapsed = time.perf_counter() - r0
        r_tflops = r_flops / (r_kernel_ms * 1e-3) / 1e12 if r_kernel_ms > 0 else 0.0
        relaxed = {"value": round(args.width * args.height * spp_total * rsteps / 1e6 / relapsed, 2), "unit": "Msamples/s", "steps": rsteps,
                   "ms_per_step": round(relapsed * 1e3 / rsteps, 3), "kernel_ms_per_launch": round(r_kernel_ms / (rsteps * max(1, rst.chunks)), 3),
                   "roofline_achieved": round(r_tflops, 3), "roofline_frac": round(r_tflops / PEAK_FP32_VALU_TFLOPS, 4),
                   "parity": "statistical image gate (relMSE < 1e-3 + z-test), not bit-exact; camera samples, wavelengths and the hit of a given ray are bit-exact",
                   "build": "libiris_hip_relaxed.so (-DIRIS_RELAXED=1)"}
        rgpu.close()

    if rank == 0:
        samples_per_step = args.width * args.height * spp_total
        value = samples_per_step * args.steps / 1e6 / elapsed
        roofline = roofline_of(iris_amd, scene, workload_key(args.scene, args.width, args.height, args.spp, opts), args.width, args.height, args.steps, st, kernel_ms, flops)
        line = {
            "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config B per GPU: %s scene %dx%d, %d spp per GPU (%d spp total), HwssNaive, "
                                   "max_depth %d, spp-split + RCCL all-reduce of the XYZ film" %
                                   ({"s1": "default (scene.rs:43-66, 3 spheres)", "s2": "10k-sphere (hierarchy unless --opt bvh_min_prims=0)", "s3": "glossy/dielectric"}[args.scene[:2]]
                                    + (" + %dx%d HDR environment map" % (args.env_size, args.env_size) if args.scene.endswith("env") else ""),
                                    args.width, args.height, args.spp, spp_total, max_depth),
                       "width": args.width, "height": args.height, "spp_per_gpu": args.spp, "parallelism": "spp-split x%d" % world,
                       "ranks": dist.get_world_size() if world > 1 else 1},
            # rank 0's path-tracing kernel, timed with HIP events on its own stream inside the library.  achieved/frac count
            # the flops of the work the kernel EXECUTES; achieved_reference_work prices everything the reference would execute
            # samples the kernel traces per second: those of pixels the sky mask proves black are counted in `value` (their result,
            # zero, is the reference's bit for bit) but cost nothing; with an environment map the mask is off and the two are equal
            "value_traced_only": round((st.samples - st.sky_samples) * world * args.steps / 1e6 / elapsed, 2),
            "roofline": roofline,
            # the second kernel of a step, HBM-bound: it reads the 12-byte result of every stored sample once and writes the film
            "roofline_resolve": {"bound": "hbm", "achieved": round(resolve_bytes / (resolve_ms * 1e-3) / 1e9, 1) if resolve_ms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(resolve_bytes / (resolve_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if resolve_ms > 0 else 0.0,
                                 "traffic": None, "kernel": "k_resolve", "kernel_ms_per_step": round(resolve_ms / args.steps, 3),
                                 "bytes_model": "12 B x stored samples (samples - sky_samples) read + 12 B x pixels written + 1 B x pixels of the sky mask"},
            # rays the kernel traces per second; the primary rays of sky-masked samples are counted by the work counters (as the
            # reference executes them) but never traced, so they are left out here and priced in the second figure only
            "rays_per_s": round((st.rays - st.sky_samples) / (kernel_ms / args.steps * 1e-3), 1) if kernel_ms else None,
            "rays_per_s_reference_work": round(st.rays / (kernel_ms / args.steps * 1e-3), 1) if kernel_ms else None,
            "counters_per_step": {"samples": st.samples, "bounces": st.bounces, "rays": st.rays, "sphere_tests": st.sphere_tests, "bvh_depth": st.bvh_depth,
                                  "sky_samples": st.sky_samples},
        }
        if config_c is not None:
            line["config_c_strong"] = config_c
        if relaxed is not None:
            line["relaxed_mode"] = relaxed
        # ---- fourth leg, one GPU, default flags only: the other BASELINE configurations, two timed steps each, so that the
        # driver's record carries them (config D: 10k spheres; config E: glossy/dielectric under the HDR sky, depth 32; S3 without
        # the sky; the default scene under the sky, where no sample is skipped)
        if world == 1 and args.scene == "s1" and not opts and not args.no_other_configs and not args.multi_kernel:
            del film
            line["other_configs"] = {
                name: other_config(iris_amd, torch, name, label, args.width, args.height, spp, 2, args.env_size, local_rank)
                for name, label, spp in (
                    ("s2", "BASELINE config D: synthetic 10k-sphere scene (hierarchy), 64 spp", 64),
                    ("s3env", "BASELINE config E: glossy/dielectric scene under the %dx%d HDR sky, 256 spp, max_depth 32" % (args.env_size, args.env_size), 256),
                    ("s3", "glossy/dielectric scene without the sky, 256 spp, max_depth 32", 256),
                    ("s1env", "default scene under the %dx%d HDR sky (BASELINE's wording of configs A-C), 256 spp" % (args.env_size, args.env_size), 256))}
        if world == 1 and not args.no_cpu_baseline:
            n_threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
            line["cpu_baseline"] = cpu_baseline(n_threads)
        print(json.dumps(line), flush=True)
    gpu.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
