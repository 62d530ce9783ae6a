#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame: BASELINE.json configs[1] = demo default sphere scene (seed 0), 1920x1080, 1 spp, 8 bounces,
Russian roulette on, sky-gradient environment, FrameIndex = step, camera jitter = Halton2D(step % 8 + 1) - 0.5.
With N > 1 the frame is tile-partitioned (32x32 tiles, interleaved; rank 0, which assembles the frame, carries a larger,
auto-tuned share because its tiles need no transfer), every rank renders its tiles, the HDR tiles are gathered to rank 0
over RCCL -- one collective per batch of frames in flight -- and un-swizzled there (strong scaling: the frame is fixed).
See directx-raytracing-spheres-demo_amd/exchange.py.
Inputs (scene, BVH) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, BOUNCES, SPP = 1920, 1080, 8, 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed frames rendered before the --warmup frames, to bring a cold GPU to its running clocks (the driver's run is ~2 ms of GPU "
                         "work in all); -1 = 256 for frames up to 4 M pixel-samples, 8 beyond; 0 = none; reported as config.prewarm_frames")
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--bounces", type=int, default=BOUNCES)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--scene", choices=["demo", "small", "procedural"], default="demo")
    ap.add_argument("--spheres", type=int, default=1 << 20, help="sphere count of the procedural scene")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=range(0, 9),
                    help="consecutive frames rotate over this many streams / output buffers so one frame's latency-bound tail overlaps the next "
                         "frames' start; 0 = auto (one GPU: 3, or 6 for frames under 1.5 M pixels; 4 when the frame is split over several)")
    ap.add_argument("--animate", action="store_true",
                    help="demo scene in motion (closed-form springs + Moon orbit, 1/60 s per frame): per-frame sphere upload + LBVH refit inside the timed region")
    ap.add_argument("--env-map", action="store_true",
                    help="light the demo scene with the lat-long environment map (MyScene.ixx:94-95; procedural HDR stand-in) instead of the sky")
    ap.add_argument("--textures", action="store_true",
                    help="demo scene with its textured objects (row N1: Alien-Metal, Moon, Earth; procedural stand-ins for the reference's image files)")
    ap.add_argument("--di", action="store_true", help="IsDIEnabled = 1 (row N4): sphere-light direct illumination pass before the bounce passes")
    ap.add_argument("--force-tiles", action="store_true", help="run the tile / gather / un-swizzle path even with one rank (rehearsal of the N > 1 path)")
    ap.add_argument("--root-weight", type=int, default=-1,
                    help="tiled path: shares of the frame rank 0 renders (every other rank renders one; 0 = rank 0 renders everything); "
                         "-1 = measure a few candidates before the warm-up and keep the fastest")
    ap.add_argument("--gather-batch", type=int, default=0, help="tiled path: frames per RCCL gather (0 = frames in flight)")
    ap.add_argument("--gather", choices=["cabi", "torch"], default="torch",
                    help="tiled path, N > 1: torch.distributed.gather (default) or the C-ABI's pt_gather (falls back to torch if its self-check fails). "
                         "With the default the C-ABI exchange is still checked and timed AFTER the measurement, under a watchdog (result: config.tile_exchange.cabi_check)")
    ap.add_argument("--moving-camera", action="store_true",
                    help="the camera position changes every frame (a slow orbit without a turn, as App::Update's camera block moves it: Source/App.cpp:531-553): the exact "
                         "primary-beam lists of a resting view do not apply; the renderer keeps lists with slack instead (PT_BEAM_REACH=0 turns those off)")
    ap.add_argument("--turning-camera", action="store_true",
                    help="the camera turns every frame (a slow yaw with a little pitch, 1.3e-4 rad per frame -- 95 degrees per second at this frame rate -- as App::Update's "
                         "mouse look turns it); with --moving-camera it travels as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-row-step", type=int, default=1, help="the CPU baseline renders every n-th row of each frame")
    args = ap.parse_args()
    # stdout carries exactly one line, the result.  Libraries print there too (RCCL announces its version on stdout when the
    # communicator comes up), so file descriptor 1 is pointed at stderr for the run and the result goes to the saved original.
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on these hosts (normally already exported)
    import torch
    import torch.distributed as dist

    import dxrs_amd_loader  # noqa: F401
    import dxrs_amd
    from dxrs_amd.types import graphics_settings

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # PT_BENCH_REHEARSAL=1: every rank on GPU 0, process group over gloo, the tile exchange staged through the host -- a rehearsal of this
    # script's N > 1 control flow (autotune, batches, barriers, totals) on a one-GPU box, where RCCL refuses ("Duplicate GPU detected").
    # Its numbers mean nothing (the ranks share one GPU) and the line says so.
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    tiled = world > 1 or args.force_tiles
    if tiled:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    host = dxrs_amd.load_host()
    kind = {"demo": dxrs_amd.host.SCENE_DEMO, "small": dxrs_amd.host.SCENE_SMALL, "procedural": dxrs_amd.host.SCENE_PROCEDURAL}[args.scene]
    spheres, materials, sd = host.scene(kind, seed=1 if args.scene == "procedural" else 0, count=args.spheres)
    w, h = args.width, args.height

    # a real (non-default) torch stream: the renderer's kernels and the RCCL gather are ordered through it
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    if args.frames_in_flight == 0:
        # 3 lanes, which the library puts on hardware queues of their own (pt_create: highest-priority pool), so that neither the
        # caller's stream nor the RCCL gather on it shares a queue with a lane.  Small untiled frames are bound by the dependent
        # chain of a frame's two launches rather than by throughput and gain from 6 lanes in the default pool, as long as the
        # process has few other streams (256x256: 0.0186 -> 0.0162 ms, 640x384: 0.0356 -> 0.0313); the tiled path is best with 3
        # (tools/experiments/lanes_prio2.sh, tilecost.py: 1080p in tiles 0.113 ms against 0.139 with 4 default-pool lanes)
        args.frames_in_flight = 3 if (world > 1 or args.force_tiles) else (6 if args.width * args.height < 1500000 else 3)
    nbuf = args.frames_in_flight
    r = dxrs_amd.Renderer(device=local_rank, stream=stream, frames_in_flight=nbuf)
    tex = None
    if args.textures or args.env_map:
        if args.scene != "demo":
            raise SystemExit("--textures / --env-map are defined for the demo scene")
        tex, sd_env = host.demo_textures(0, 0.0, textured=args.textures, environment_map=args.env_map, return_scene_data=True)
        if args.env_map:
            sd = sd_env  # names the environment map in the texture table
    accel = r.set_scene(spheres, materials, sd)
    first_build_ms = float(accel.build_ms)   # includes the one-time code-object load of the sort kernels
    accel = r.build_accel()                  # steady-state full rebuild (what a per-frame TLAS rebuild would cost)
    if tex is not None:
        r.set_textures(tex)
    gs = graphics_settings(w, h, frame_index=0, bounces=args.bounces, spp=args.spp, di=args.di)
    r.set_constants(gs)
    cams = [host.camera(w, h, jitter_index=k, jitter_count=8) for k in range(8)]
    if args.moving_camera or args.turning_camera:
        import math
        n_cam = 3 * args.steps + args.warmup + 64
        where = (lambda k: (0.6 * math.sin(0.01 * k), 0.05 * math.sin(0.013 * k), -15.0 + 0.4 * math.cos(0.01 * k))) if args.moving_camera else (lambda k: (0.0, 0.0, -15.0))
        # (look_at = None: the demo camera's fixed orientation; turning: the point looked at wanders, 15 units ahead)
        aim = (lambda k: (2.0 * math.sin(0.001 * k), 0.5 * math.sin(0.0013 * k), 0.0)) if args.turning_camera else (lambda k: None)
        cams = [host.camera(w, h, position=where(k), look_at=aim(k), jitter_index=k % 8, jitter_count=8) for k in range(n_cam)]

    def set_frame(k, rr=None):
        rr = rr or r
        gs.FrameIndex = k
        if args.animate:
            rr.update_spheres(anim[k])  # upload + LBVH refit on this frame's stream (Scene::Refresh + TLAS rebuild analogue)
        rr.set_camera(cams[k % len(cams)])
        rr.set_constants(gs)

    # multi-buffered outputs: frame k writes buffer k % frames_in_flight (the reference's swap chain, generalised)
    tune_log = {}
    cabi, ops, ex = False, None, None
    if not tiled:
        frames = [torch.empty((h * w, 4), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    else:
        from dxrs_amd.exchange import HipOps, TileExchange
        # the batch must cover the frames in flight: a batch buffer is reused two batches later, and a frame only waits for
        # the caller-stream marker frames_in_flight - 1 calls back (see pt_api.hip render_common)
        batch = max(args.gather_batch or nbuf, nbuf - 1, 1)
        # The exchange itself: the C-ABI's pt_gather (RCCL grouped send/recv behind include/pt_api.h -- what a C++ host uses),
        # verified on the live job with a known pattern before it is trusted; torch.distributed.gather otherwise.
        if args.gather == "cabi" and world > 1 and not rehearsal:
            cabi = init_cabi_gather(r, dist, torch, dev, rank, world)
        ops = HipOps(r, dev, set_frame, cabi_gather=cabi)
        if rehearsal and world > 1:
            ops.gather_parts = lambda send, recv, nbytes: staged_gather(torch, dist, rank, world, send, recv, nbytes)
        ex = TileExchange(ops, w, h, rank, world, batch, rehearse=args.force_tiles and world == 1, frames_in_flight=nbuf)
        gather_kind = ("gloo, staged through the host (PT_BENCH_REHEARSAL: all ranks on one GPU, timings meaningless)" if rehearsal else
                       "pt_gather (C-ABI, RCCL send/recv)" if cabi else "torch.distributed.gather (RCCL)")
    if args.animate:
        if args.scene != "demo":
            raise SystemExit("--animate is defined for the demo scene")
        n_anim = 2 * args.steps + args.warmup + 64  # (timed region, event-bracketed region, the one-frame-at-a-time frames)
        anim = [host.scene_at_time(0, k / 60.0) for k in range(n_anim)]  # host-side Tick (MyScene::SetTime), precomputed

    def step(k):
        if not tiled:
            set_frame(k)
            r.render_device(frames[k % nbuf].data_ptr())
        else:
            ex.submit(k)  # render this rank's tiles; every `batch` frames: one RCCL gather + un-swizzle on rank 0

    def step_on(rr, k):
        """one frame on renderer rr (this rank's share, no gather): used for the exclusive-kernel measurement"""
        set_frame(k, rr)
        if not tiled:
            rr.render_device(frames[0].data_ptr())
        elif ex.own_px:
            rr.render_tiles(ex.own[0][0].data_ptr())

    def sync_all():
        torch.cuda.synchronize(dev)
        if tiled:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_steps(first, n):
        """n frames, fully drained: barrier + synchronize on both sides; returns seconds"""
        sync_all()
        t = time.perf_counter()
        for k in range(n):
            step(first + k)
        if tiled:
            ex.finish()  # flush a partial last batch
        torch.cuda.synchronize(dev)
        if tiled:
            dist.barrier()
            torch.cuda.synchronize(dev)
        return time.perf_counter() - t

    if tiled and world > 1:
        if args.root_weight >= 0:
            ex.configure(args.root_weight)
        else:
            # untimed, before the warm-up: try a few root weights on the live job and keep the fastest
            # (five candidates x two runs of two batches: a few dozen frames, small against the measurement itself)
            n_tune = 2 * ex.batch
            ex.autotune(lambda e: run_steps(0, n_tune), sync_all, candidates=[1, 2, 4, 8, 0], log=tune_log)
    # Untimed, before the warm-up proper: a freshly leased GPU takes tens of milliseconds of work to reach the state a renderer runs in
    # (tools/experiments/warm20.sh: the same 20 timed frames take 0.091-0.100 ms each after 5 frames, 0.087-0.092 after 3000), and the
    # contract's region is 20 frames after 5.  The same frames as the warm-up, over and over; every rank renders the same number.
    prewarm = 0 if rehearsal else args.prewarm if args.prewarm >= 0 else (256 if w * h * args.spp <= 4200000 else 8)  # (a rehearsal's timings mean nothing)
    if prewarm:
        sync_all()
        n_cycle = max(1, min(args.warmup, 32))
        for k in range(prewarm):
            step(k % n_cycle)
        if tiled:
            ex.finish()
        sync_all()
    run_steps(0, args.warmup)
    r.totals(reset=True)
    elapsed = run_steps(args.warmup, args.steps)
    tot = r.totals(reset=True)
    queue_sizes = r.queue_sizes()
    # Second timed region, identical except that a HIP event pair brackets every kernel launch on the stream it runs on:
    # it provides the per-launch durations of the roofline object.  (The event records cost ~10 % of frame time at this
    # frame size, which is why `value` comes from the first region.)
    prof, elapsed_ev = None, None
    if not args.no_roofline:
        r.set_profiling(True)
        elapsed_ev = run_steps(args.warmup + args.steps, args.steps)
        prof = r.profile(reset=True)
        r.set_profiling(False)
        tot_ev = r.totals(reset=True)
    rays, paths = int(tot.rays), int(tot.paths)
    # The reference's own frame contract: one frame at a time -- Tick -> Render -> WaitForGPU (Source/App.cpp:144-186).  `value` above is
    # the throughput of frames in flight (its swap chain, generalised); this is the latency of a frame the caller waits for.
    serial = None
    if not tiled:
        k0 = args.warmup + 2 * args.steps
        lat = []
        for k in range(24):
            set_frame(k0 + k)
            torch.cuda.synchronize(dev)
            t0s = time.perf_counter()
            r.render_device(frames[k % nbuf].data_ptr())
            torch.cuda.synchronize(dev)
            lat.append(time.perf_counter() - t0s)
        lat = sorted(lat[4:])
        serial = {"ms_per_frame": sum(lat) / len(lat) * 1e3, "median_ms": lat[len(lat) // 2] * 1e3, "min_ms": lat[0] * 1e3, "frames": len(lat),
                  "contract": "one frame at a time: submit, then wait for the GPU (App::Tick -> Render -> WaitForGPU, Source/App.cpp:144-186); host submit + sync included"}
        r.totals(reset=True)
    if tiled:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rays, paths], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays, paths = int(c[0].item()), int(c[1].item())
        # what every rank did, as seen by rank 0: a rank that rendered nothing, or a world that is not the one asked for, shows here
        mine = torch.tensor([int(tot.rays), int(tot.pixels)], dtype=torch.int64, device=dev)
        per_rank = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = [[int(v) for v in t_.tolist()] for t_ in per_rank]

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        result = {
            "metric": f"Mrays/s (CastRay-equivalents incl. primaries) at {w}x{h}, {args.bounces} bounces, {args.spp} spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "frames_per_s": args.steps / elapsed,
            "mpaths_per_s": paths / elapsed / 1e6,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "latency_ms_one_frame": serial["ms_per_frame"] if serial else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene} sphere scene (seed {1 if args.scene == 'procedural' else 0}, {len(spheres)} spheres), {w}x{h}, {args.spp} spp, {args.bounces} bounces, RR on, " + ("lat-long HDR environment map (procedural stand-in, 1024x512)" if args.env_map else "sky env")
                            + (", textured (Alien-Metal, Moon, Earth; procedural stand-in images)" if args.textures else "")
                            + (", sphere-light direct illumination (IsDIEnabled)" if args.di else "")
                            + (f", 32x32 tiles interleaved over {world} GPU(s) + RCCL gather to rank 0" if tiled else ""),
                **({"tile_exchange": {"gather": gather_kind, "root_weight": ex.root_weight, "frames_per_gather": ex.batch, "root_tiles": ex.n_root,
                                      "tiles_per_other_rank": ex.n_other, "autotune": tune_log or None,
                                      "ranks_seen": world, "rays_and_pixels_per_rank": per_rank}} if tiled else {}),
                **({"rehearsal": "PT_BENCH_REHEARSAL=1: all ranks on ONE GPU over gloo -- control-flow rehearsal, not a measurement"} if rehearsal else {}),
                "frames_in_flight": args.frames_in_flight,
                "prewarm_frames": prewarm,
                "one_frame_at_a_time": serial,
                "animated": bool(args.animate),
                "moving_camera": bool(args.moving_camera), "turning_camera": bool(args.turning_camera),
                # the primary pass of a RESTING view takes its candidates from cached per-block sphere lists (DESIGN.md "Primary beams"); a moving
                # camera or an animated scene traverses per ray -- run with --moving-camera / --animate for those figures
                "primary_beams": {"frames_using_cached_lists": int(tot.beams_used), "of": args.steps},
                "rays_per_frame": rays / args.steps,
                "accel": {"builder": {0: "device LBVH (Morton order, Karras)", 1: "host LBVH", 2: "host SAH topology (scenes up to 4096 spheres) + device boxes"}.get(int(accel.builder), str(int(accel.builder))),
                          "built": "before the timed region; --animate refits it inside", "nodes": int(accel.node_count), "depth": int(accel.depth),
                          "lds_resident": bool(accel.lds_resident), "build_ms": float(accel.build_ms), "first_build_ms": first_build_ms},
            },
        }

    # ---- roofline of the dominant kernel class, from the per-launch HIP events of the timed region itself (this rank)
    if not args.no_roofline and rank == 0:
        result["roofline"] = roofline_object(args, w, h, world, tiled, prof, tot_ev, queue_sizes, elapsed, elapsed_ev, ms_per_step, len(spheres),
                                             lambda: exclusive_profile(dxrs_amd, local_rank, stream, spheres, materials, sd, gs, tex, tiled, ex if tiled else None, step_on, args))

    # ---- CPU baseline: the scalar oracle on this node's host cores, bounded sample of the same frame (rank 0, N = 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import load_oracle

        oracle = load_oracle()
        cores = min(os.cpu_count() or 1, 16)  # a one-GPU box's share of the host (the node has 256 hardware threads for 8 GPUs)
        # bounded sample: whole frames of the same workload until ~15 s of CPU work (threads x wall) or 64 frames
        o_rays, o_time, o_frames = 0, 0.0, 0
        while o_frames < 64 and o_time * cores < 15.0:
            gs.FrameIndex = args.warmup + o_frames
            t0c = time.perf_counter()
            _, ost = oracle.render(spheres, materials, sd, cams[(args.warmup + o_frames) % len(cams)], gs, row_step=args.cpu_row_step, threads=cores, textures=tex)
            o_time += time.perf_counter() - t0c
            o_rays += int(ost.rays)
            o_frames += 1
        # single-thread figure: every 16th row of one frame
        t0c = time.perf_counter()
        _, ost1 = oracle.render(spheres, materials, sd, cams[args.warmup % len(cams)], gs, row_step=16, threads=1, textures=tex)
        t_single = time.perf_counter() - t0c
        result["cpu_baseline"] = {
            "value": o_rays / o_time / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "single_thread_value": ost1.rays / t_single / 1e6,
            "sample": f"{o_frames} frame(s) of the same {w}x{h} workload (FrameIndex {args.warmup}..{args.warmup + o_frames - 1}"
                      + (f", every {args.cpu_row_step}th row" if args.cpu_row_step > 1 else "")
                      + f"): {o_rays} rays in {o_time:.2f} s wall = {o_time * cores:.1f} s of CPU work, scalar C oracle, "
                      + ("its own median-split BVH" if len(spheres) > 64 else "brute-force O(N) intersection")
                      + f" over {len(spheres)} spheres, {cores} threads",
        }

    if rank == 0:
        os.write(_RESULT_FD, (json.dumps(result) + "\n").encode())  # the ONE line of this job's stdout
    # The C-ABI's own exchange (pt_comm_init / pt_gather) has never met more than one GPU before the first multi-GPU run of this
    # script, so the measurement above went through torch.distributed.gather.  With the result line safely out, pt_gather is now
    # brought up, checked bit for bit against the frames of the torch path and timed -- under a watchdog, so that a rank stuck in
    # ncclCommInitRank ends the job instead of hanging it.  Outcome: one tagged line on stderr (+ gpurun_out/cabi_check_<N>gpus.json).
    if tiled and world > 1 and not rehearsal and not cabi:
        cabi_post_check(r, ex, ops, dist, torch, dev, rank, world, run_steps, args)
    r.close()
    if tiled:
        dist.destroy_process_group()


def cabi_post_check(r, ex, ops, dist, torch, dev, rank, world, run_steps, args):
    import threading

    def give_up():
        if rank == 0:
            print('[bench] cabi_check: {"ok": false, "reason": "timed out after 60 s (a rank did not come back from pt_comm_init / pt_gather)"}', file=sys.stderr, flush=True)
        os._exit(0)  # the result line is already out; a hung collective cannot be unwound

    dog = threading.Timer(60.0, give_up)
    dog.daemon = True
    dog.start()
    out = {"ok": False}
    try:
        def batch_frames(first):
            for k in range(ex.batch):
                ex.submit(first + k)
            ex.finish()
            torch.cuda.synchronize(dev)
            dist.barrier()
            return [f.clone() for f in ex.frames[: ex.batch]] if rank == 0 else None

        first = args.warmup + 3 * args.steps + 64
        ref = batch_frames(first)
        t_torch = run_steps(first, 2 * ex.batch) / (2 * ex.batch)
        if init_cabi_gather(r, dist, torch, dev, rank, world):
            ops.gather_parts = ops._gather_parts
            got = batch_frames(first)
            same = 1
            if rank == 0:
                same = int(all(torch.equal(a, b) for a, b in zip(ref, got)))
            flag = torch.tensor([same], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            t_cabi = run_steps(first, 2 * ex.batch) / (2 * ex.batch)
            out = {"ok": bool(int(flag.item())), "frames_bit_identical_to_torch_gather": bool(int(flag.item())),
                   "ms_per_step_pt_gather": t_cabi * 1e3, "ms_per_step_torch_gather": t_torch * 1e3, "frames_compared": ex.batch}
        else:
            out = {"ok": False, "reason": "pt_comm_init or its self-check failed on some rank (see stderr)"}
    except Exception as e:  # noqa: BLE001
        out = {"ok": False, "reason": f"{type(e).__name__}: {e}"}
    dog.cancel()
    if rank == 0:
        print("[bench] cabi_check: " + json.dumps(out), file=sys.stderr, flush=True)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"cabi_check_{world}gpus.json"), "w") as f:
                json.dump(out, f)
        except OSError:
            pass


def load_counters(workload):
    """profiles/counters_<workload>.json (rocprofv3 PMC passes, made by profiles/collect.sh) -- only if it was measured on the
    kernel sources this run uses; otherwise (None, reason)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import kernel_knobs, kernel_source_hash
    path = os.path.join(ROOT, "profiles", f"counters_{workload}.json")
    try:
        c = json.load(open(path))
    except Exception:
        return None, "no committed counters for this workload"
    if c.get("workload") != workload:
        return None, "committed counters are for another workload"
    if c.get("kernel_source_hash") != kernel_source_hash():
        return None, f"committed counters are stale (measured on kernel sources {c.get('kernel_source_hash')}, running {kernel_source_hash()})"
    if c.get("knobs", {}) != kernel_knobs():
        return None, f"committed counters were taken under other PT_* knobs ({c.get('knobs', {})}) than this run's ({kernel_knobs()})"
    return c, None


def exclusive_profile(dxrs_amd, local_rank, stream, spheres, materials, sd, gs, tex, tiled, ex, step_on, args):
    """the same frames one at a time (one frame in flight, 20 frames): per-launch durations without other frames sharing the GPU"""
    r1 = dxrs_amd.Renderer(device=local_rank, stream=stream, frames_in_flight=1)
    r1.set_scene(spheres, materials, sd); r1.set_constants(gs)
    if tex is not None:
        r1.set_textures(tex)
    if tiled:
        r1.set_partition_ex(*ex.range)
    for k in range(4):
        step_on(r1, k)
    r1.set_profiling(True)
    for k in range(20):
        step_on(r1, args.warmup + k)
    p1 = r1.profile(reset=True)
    r1.close()
    return p1


def roofline_object(args, w, h, world, tiled, prof, tot_ev, qs, elapsed, elapsed_ev, ms_per_step, n_spheres, exclusive):
    """`achieved` = algorithmic bytes of the dominant kernel class per launch / its average launch duration (HIP events of
    this run); algorithmic bytes = 160 B per ray traced + 40 B per path finished by those launches, plus -- for scenes traversed in
    global memory (C5) -- the scene term from the kernels' own counters: 64 B per node record read + 16 B per sphere record tested."""
    n_f = args.steps
    my_pixels = tot_ev.pixels / n_f
    my_rays = tot_ev.rays / n_f
    my_secondary = my_rays - my_pixels
    split = prof.shade_launches > 0
    workload = f"{args.scene}-{w}x{h}-{args.spp}spp-{args.bounces}b"
    plain = not (args.di or args.textures or args.env_map or args.animate or tiled)
    counters, stale = load_counters(workload) if plain else (None, "counters are collected for the plain single-GPU workloads only")
    # `bound` / `achieved` / `frac` are the measurement contract's HBM figures in SURVEY 8(d)'s NOTIONAL bytes; `binding` names what the
    # kernels are really limited by (VALU issue: see `valu` and `traffic`)
    out = {"bound": "hbm", "binding": "valu", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    if split:
        # split schedule (BVH in global memory): every ray is traced by a traversal-type launch (primary_kernel, traverse[_dyn]_kernel,
        # tail_kernel); the class is all of them
        name, cls = "traversal kernels of the split schedule (primary_kernel + traverse_dyn_kernel + tail_kernel)", ("primary", "traverse", "tail")
        # SURVEY 8(d) prices a node visit at 32 B (a box-in-own-node record); the records this build reads are 64 B (two child boxes:
        # one fetch tests both children).  `frac` uses the bytes really read; `frac_survey_32B_nodes` is the contract's literal figure.
        scene_bytes = 64.0 * tot_ev.node_visits + 16.0 * tot_ev.sphere_tests
        scene_bytes_32 = 32.0 * tot_ev.node_visits + 16.0 * tot_ev.sphere_tests
        b = (160.0 * my_rays + 40.0 * tot_ev.paths / n_f) * n_f + scene_bytes
        ms, n = prof.ms_traverse + prof.ms_tail, prof.traverse_launches + prof.tail_launches
        out["scene_term"] = {"node_visits_per_ray": tot_ev.node_visits / max(tot_ev.rays, 1), "sphere_tests_per_ray": tot_ev.sphere_tests / max(tot_ev.rays, 1),
                             "bytes_per_frame": scene_bytes / n_f, "accounting": "64 B per node record read + 16 B per sphere record tested (device counters of the traversal kernels)"}
        impl = None
        rays_class = my_rays
    else:
        n_wf = prof.traverse_launches // n_f  # compacting passes per frame (incl. the primary pass)
        inline1 = tot_ev.rays_first_pass_inline / n_f  # bounce-1 rays the primary pass traced in registers (never queued)
        if qs and len(qs) > n_wf >= 1:
            wf_in, wf_out, loop_in = sum(qs[1:n_wf]), sum(qs[1:n_wf + 1]), qs[n_wf]
            px_wf = qs[0] - loop_in if world == 1 else my_pixels - loop_in
        else:  # no per-queue sizes (spp > 1): every secondary ray is written once and read once
            wf_in = wf_out = my_secondary; loop_in = 0; px_wf = my_pixels
        rays_wf = my_pixels + inline1 + wf_in          # compacting passes: primaries, in-register bounce-1 rays, queued rays
        rays_loop = max(my_rays - rays_wf, 0.0)        # everything else is traced by the looping pass
        if prof.ms_traverse >= prof.ms_tail:
            name, cls = "bounce_kernel<primary> (primary / compacting trace+shade passes)", ("bounce<primary>", "bounce<compact>")
            b, ms, n, rays_class = (160.0 * rays_wf + 40.0 * px_wf * args.spp) * n_f, prof.ms_traverse, prof.traverse_launches, rays_wf
            impl = (48.0 * wf_in + 48.0 * wf_out + 16.0 * px_wf) * n_f
        else:
            name, cls = "bounce_kernel<loop> (looping trace+shade pass)", ("bounce<loop>",)
            b, ms, n, rays_class = (160.0 * rays_loop + 40.0 * loop_in * args.spp) * n_f, prof.ms_tail, prof.tail_launches, rays_loop
            impl = (48.0 * loop_in + 16.0 * loop_in) * n_f
    achieved = (b / n) / (ms / n * 1e-3) / 1e9 if n and ms > 0 else 0.0
    if split and n and ms > 0:
        out["frac_survey_32B_nodes"] = ((b - scene_bytes + scene_bytes_32) / n) / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS
    out.update(kernel=name, achieved=achieved, frac=achieved / HBM_PEAK_GBS, bytes_per_launch=b / max(n, 1), avg_launch_ms=ms / max(n, 1), launches_per_frame=n / n_f,
               rays_per_launch=rays_class / max(n / n_f, 1),
               accounting="SURVEY 8(d): 160 B per ray traced + 40 B per path finished by these launches" + (" + the scene term" if split else ""))
    # measured HBM traffic of the class (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per launch, committed by profiles/collect.sh)
    traffic = None
    if counters:
        per = [counters["kernels"][k] for k in cls if k in counters["kernels"] and "hbm_bytes_per_launch" in counters["kernels"][k]]
        if per:
            traffic = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in per) / sum(k["launches"] for k in per)
    out["traffic"] = traffic
    out["counters"] = {"file": f"profiles/counters_{workload}.json", "used": counters is not None, "reason": stale}
    if impl is not None:
        out["implementation_bytes_per_launch"] = impl / max(n, 1)  # what this fused implementation has to move: rays kept in registers cost nothing
    # What actually binds: VALU issue.  gfx950 issues one wave64 VALU instruction per SIMD every 4 cycles, or two in one 4-cycle slot when
    # they are adjacent, independent and at least one is an fp32 fma / mul / add (profiles/r02_valu_rate.txt, r02_valu_mix.txt); this
    # branchy scalar code pairs little, so both bounds are given.
    if counters and all("valu_insts_per_launch" in k for k in counters["kernels"].values()):
        per_frame = sum(k["valu_insts_per_launch"] * k["launches"] for k in counters["kernels"].values()) / counters["frames_per_pass"]
        t4, t2 = per_frame * 4 / 1024 / 2.4e9 * 1e3, per_frame * 2 / 1024 / 2.4e9 * 1e3
        out["valu"] = {"wave_insts_per_frame": per_frame, "bound_ms_unpaired_4_cycles": t4, "bound_ms_fully_paired_2_cycles": t2,
                       "frac_of_unpaired_bound": t4 / ms_per_step, "frac_of_paired_bound": t2 / ms_per_step,
                       "wait_fractions": {k: {f: v[f] for f in ("active_frac", "wait_any_frac", "wait_inst_frac") if f in v} for k, v in counters["kernels"].items()},
                       "source": "SQ_INSTS_VALU per launch (rocprofv3, one frame in flight) x 4 or 2 cycles / 1024 SIMDs / 2.4 GHz"}
    else:
        out["valu"] = None
    out.update(compacting_ms_per_frame=prof.ms_traverse / n_f, shade_ms_per_frame=prof.ms_shade / n_f, loop_ms_per_frame=prof.ms_tail / n_f,
               ms_per_step_with_events=elapsed_ev / n_f * 1e3, sustained_gbs=(b / n_f) / (elapsed / args.steps) / 1e9)
    excl = None
    if args.frames_in_flight > 1 and not split:
        p1 = exclusive()
        ms1, n1 = (p1.ms_tail, p1.tail_launches) if "loop" in name else (p1.ms_traverse, p1.traverse_launches)
        if n1 and ms1 > 0:
            excl = {"avg_launch_ms": ms1 / n1, "achieved": (b / n) / (ms1 / n1 * 1e-3) / 1e9}
            excl["frac"] = excl["achieved"] / HBM_PEAK_GBS
    out["exclusive"] = excl
    out["note"] = ("bound / achieved / frac follow the measurement contract: NOTIONAL wavefront bytes (SURVEY 8(d)) over the launch duration, not a bandwidth -- "
                   "'traffic' is the measured HBM bytes per launch (far below: the fused kernels keep rays in registers), 'valu' the instruction-issue bound "
                   "that binds; with N frames in flight launches of consecutive frames overlap and stretch ('exclusive' = one frame at a time)")
    return out


def staged_gather(torch, dist, rank, world, send, recv, nbytes):
    """pt_gather's contract (include/pt_api.h: every rank but the root sends nbytes, rank r's part lands at recv + (r - 1) * nbytes, the
    root contributes nothing) over gloo with host staging -- PT_BENCH_REHEARSAL only."""
    mine = torch.zeros(nbytes, dtype=torch.uint8) if rank == 0 else send.contiguous().view(torch.uint8).reshape(-1)[:nbytes].cpu()
    parts = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank == 0:
        store = torch.empty(0, dtype=torch.uint8, device=recv.device).set_(recv.untyped_storage())
        base = recv.storage_offset() * recv.element_size()
        for q in range(1, world):
            store[base + (q - 1) * nbytes: base + q * nbytes].copy_(parts[q])


def init_cabi_gather(r, dist, torch, dev, rank, world):
    """pt_comm_init on every rank (id from rank 0 through the torch process group) + a self-check gather of a known pattern.
    Returns True when every rank saw it work; any failure is symmetric (all ranks fall back together)."""
    # every rank first shows it can load RCCL at all (a rank that cannot must not leave the others waiting in ncclCommInitRank)
    try:
        my_id, can = r.comm_unique_id(), 1
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {rank}: RCCL cannot be loaded through the C-ABI ({e}); using torch.distributed.gather", file=sys.stderr)
        my_id, can = None, 0
    flag = torch.tensor([can], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return False
    ok = 1
    ids = [my_id if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    try:
        r.comm_init(ids[0], rank, world)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {rank}: pt_comm_init failed ({e}); using torch.distributed.gather", file=sys.stderr)
        ok = 0
    # "communicator up" is agreed on BEFORE the first pt_gather: a rank whose pt_comm_init failed must not leave the others in the receive group
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        try:
            r.comm_destroy()
        except Exception:  # noqa: BLE001
            pass
        return False
    try:
        n = 4096
        send = torch.full((n,), float(rank), dtype=torch.float32, device=dev)
        recv = torch.zeros((max(world - 1, 1), n), dtype=torch.float32, device=dev)
        r.gather(send.data_ptr() if rank else 0, recv.data_ptr() if rank == 0 else 0, 4 * n, 0)
        torch.cuda.synchronize(dev)
        if rank == 0:
            want = torch.arange(1, world, dtype=torch.float32, device=dev).view(-1, 1).expand(world - 1, n)
            ok = int(torch.equal(recv, want))
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {rank}: C-ABI gather unavailable ({e}); using torch.distributed.gather", file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        try:
            r.comm_destroy()
        except Exception:  # noqa: BLE001
            pass
        return False
    return True


if __name__ == "__main__":
    main()
