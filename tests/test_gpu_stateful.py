"""Stateful fuzzing of the C-ABI: ONE context with three frames in flight lives through a random sequence of state changes -- new
scenes (incl. empty and single-sphere ones and ones whose BVH lives in global memory), moved spheres with and without a refit,
texture tables set / replaced / dropped, object rotations, partitions, rectangles, sample counts, direct illumination, resting and
moving views (primary-beam lists built, used, invalidated) -- and EVERY frame it renders is compared with the CPU oracle bit for bit,
ray count included.  The single-shot parity tests start from a fresh state each time; this one is after what survives between calls
(per-lane scene copies, generation counters, pending uploads, cached lists): the kind of bug the round's rotation-generation defect was."""
import copy
import os

import numpy as np
import pytest

from test_gpu_fuzz import random_scene

pytestmark = pytest.mark.gpu


class _named:
    """names the call in the failure when the C-ABI accepts what it should refuse"""

    def __init__(self, what):
        self.what = what

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            raise AssertionError(self.what + ": accepted")
        return False


def _texture_set(dxrs, rng, n, with_env):
    from dxrs_amd import abi_types as A
    from dxrs_amd import textures as T
    ts = T.TextureSet(n)
    odd = rng.integers(0, 3)  # sizes the samplers' wrap addressing has to cope with: 1 x 1, odd, wide
    imgs = [ts.add_image(T.checker(*((1, 1, 1), (7, 5, 2), (32, 16, 4))[odd][:2], cells=((1, 1, 1), (7, 5, 2), (32, 16, 4))[odd][2]), srgb=True),
            ts.add_image(T.planet_albedo(64, 32, seed=int(rng.integers(0, 100))), srgb=True),
            ts.add_image(T.normal_map_from_height(T.value_noise(32, 32, seed=int(rng.integers(0, 100)))), srgb=False)]
    for i in range(n):
        if rng.random() < 0.5:
            ts.assign(i, A.TEXTURE_MAP_BASE_COLOR, imgs[int(rng.integers(0, 2))])
        if rng.random() < 0.3:
            ts.assign(i, A.TEXTURE_MAP_NORMAL, imgs[2])
        if rng.random() < 0.5:
            ts.set_rotation(i, T.quaternion_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6.28))))
    env = None
    if with_env:
        env = ts.add_hdr_image(T.sky_latlong(64, 32, seed=int(rng.integers(0, 100))))
    return ts, env


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_STATE_SEEDS", "18"))))
def test_random_call_sequences_match_oracle(dxrs, host, oracle, seed):
    import torch
    from dxrs_amd import tiles
    rng = np.random.default_rng(7000 + seed)
    sd0 = host.scene(dxrs.host.SCENE_SMALL)[2]
    tstream = torch.cuda.Stream()
    lanes = (3, 1, 2)[seed % 3]
    # the context-wide switches, too: fused / split schedule, SAH topology / device LBVH, BVH in LDS / in global memory
    from dxrs_amd import abi_types as A
    flags = (0, 0, A.PT_FLAG_SPLIT_KERNELS, A.PT_FLAG_FAST_BUILD, A.PT_FLAG_NO_LDS_SCENE, A.PT_FLAG_NO_LDS_SCENE | A.PT_FLAG_FAST_BUILD)[(seed // 3) % 6]
    r = dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=lanes, flags=flags)
    state = dict(spheres=None, materials=None, sd=None, tex=None, w=64, h=48, spp=1, bounces=4, di=False, cam_seed=0, pos=(0.0, 0.5, -12.0), frame=0, vel=None, orbit=False)
    checked = 0
    log = []
    expected = dict(rays=0)  # what the device-accumulated ray total must read (checked and reset around every tile frame and at the end)

    def check_totals():
        tot = r.totals(reset=True)
        assert tot.rays == expected["rays"], f"device ray total {tot.rays} != {expected['rays']} summed over the oracle's frames; " + " | ".join(log[-12:])
        expected["rays"] = 0

    def new_scene():
        n = int(rng.choice([0, 1, 2, 5, 40, 300, 5000]))
        if n == 0:
            s, m = random_scene(dxrs, rng, 1)
            s, m = s[:0], m[:0]
        else:
            s, m = random_scene(dxrs, rng, n)
        sd = copy.copy(sd0)
        state.update(spheres=s, materials=m, sd=sd, tex=None)
        log.append(f"scene n={len(s)}")
        r.set_scene(s, m, sd)

    def new_textures():
        n = len(state["spheres"])
        if rng.random() < 0.25:
            log.append("drop textures")
            state["tex"] = None
            sd = copy.copy(sd0)
            state["sd"] = sd
            r.set_scene(state["spheres"], state["materials"], sd)  # (a fresh pt_set_scene is how a table is dropped)
            return
        ts, env = _texture_set(dxrs, rng, n, with_env=rng.random() < 0.5)
        sd = copy.copy(sd0)
        if env is not None:
            sd.EnvironmentLightTextureDescriptor, sd.IsEnvironmentLightTextureCubeMap = env, 0
        state.update(tex=ts, sd=sd)
        log.append(f"textures env={env}")
        r.set_scene(state["spheres"], state["materials"], sd)
        r.set_textures(ts)

    def move(refit):
        s = state["spheres"].copy()
        state["prev"] = state["spheres"]
        if len(s) and refit:
            k = rng.random(len(s)) < 0.7
            s["cx"][k] += rng.uniform(-0.4, 0.4, int(k.sum())).astype(np.float32)
            s["cy"][k] += rng.uniform(-0.4, 0.4, int(k.sum())).astype(np.float32)
            s["r"][k] *= rng.uniform(0.8, 1.25, int(k.sum())).astype(np.float32)
        state["spheres"] = s
        log.append(f"move refit={refit}")
        r.update_spheres(s, refit=refit)  # without a refit the spheres are the old ones: the old boxes stay valid

    def rotate():
        if state["tex"] is None or len(state["spheres"]) == 0:
            return
        from dxrs_amd import textures as T
        for i in range(len(state["spheres"])):
            if rng.random() < 0.5:
                state["tex"].set_rotation(i, T.quaternion_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6.28))))
        log.append("rotate")
        r.update_rotations(state["tex"].rotations)

    def rejected_call():
        """a call the C-ABI must refuse -- and must leave no trace of: the frames after it still match the oracle"""
        from dxrs_amd import textures as T
        kind = int(rng.integers(0, 6))
        n = len(state["spheres"])
        if kind == 3 and n == 0:
            kind = 4  # (an empty scene has no object whose maps could name a texture)
        log.append(f"rejected call {kind}")
        with pytest.raises(dxrs.PtError), _named(f"rejected call {kind} with n = {n}, textured = {state['tex'] is not None}"):
            if kind == 0:    # the sphere count must not change
                bad = np.concatenate([state["spheres"], state["spheres"][:1]]) if n else random_scene(dxrs, rng, 1)[0]
                r.update_spheres(bad)
            elif kind == 1:  # a non-finite sphere
                bad = state["spheres"].copy() if n else random_scene(dxrs, rng, 1)[0]
                bad["cx"][0] = np.nan
                (r.update_spheres if n else (lambda b: r.set_scene(b, random_scene(dxrs, rng, 1)[1], state["sd"])))(bad)
            elif kind == 2:  # a rectangle outside the frame
                gs = dxrs.types.graphics_settings(state["w"], state["h"], bounces=1, spp=1)
                r.set_constants(gs); r.set_camera(host.camera(state["w"], state["h"]))
                r.render((state["w"] - 8, 0, 16, 8))
            elif kind == 3:  # a texture descriptor outside the table: the table in use must survive
                ts = T.TextureSet(max(n, 1))
                ts.add_image(T.checker(8, 8, cells=2), srgb=True)
                ts.assign(0, 0, 7)
                r.set_textures(ts)
            elif kind == 4:  # a partition that is none
                r.set_partition_ex(2, 2, 3)
            else:            # rotations for a different number of objects (or for an untextured scene)
                r.update_rotations(np.tile(np.array([0, 0, 0, 1], dtype=np.float32), (n + 2, 1)))

    def render():
        nonlocal checked
        w, h = state["w"], state["h"]
        gs = dxrs.types.graphics_settings(w, h, frame_index=state["frame"], bounces=state["bounces"], spp=state["spp"], di=state["di"])
        if state["vel"] is not None:  # drifting: the camera translates a little every frame and does not turn (primary-beam lists with slack)
            state["pos"] = tuple(float(np.float32(p + v)) for p, v in zip(state["pos"], state["vel"]))
        # (drifting with the eye on the origin = a slow turn on top of the travel: lists with slack AND a pixel margin; else the orientation stays)
        cam = host.camera(w, h, position=state["pos"], look_at=None if (state["vel"] is not None and not state["orbit"]) else (0.0, 0.0, 0.0), jitter_index=state["cam_seed"])
        state["frame"] += 1
        r.set_camera(cam); r.set_constants(gs)
        mode = rng.integers(0, 4)
        ref_full = None
        log.append(f"render mode={mode} {w}x{h} spp={state['spp']} b={state['bounces']} di={state['di']} tex={state['tex'] is not None} n={len(state['spheres'])}")

        def oracle_frame(rect=None):
            return oracle.render(state["spheres"], state["materials"], state["sd"], cam, gs, rect=rect, threads=8, textures=state["tex"])
        if mode == 0 and w >= 48 and h >= 40:   # a rectangle
            rect = (int(rng.integers(0, w - 40)), int(rng.integers(0, h - 32)), 40, 32)
            img, st = r.render(rect)
            ref, ost = oracle_frame(rect)
            assert st.rays == ost.rays and np.array_equal(img.view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"step {checked}: rect; " + " | ".join(log[-12:])
            expected["rays"] += ost.rays
        elif mode == 1:                          # this rank's tiles of a 3-rank job
            rank = int(rng.integers(0, 3))
            check_totals()
            r.set_partition(rank, 3)
            # (torch.empty, not zeros: a fill queued on the caller's stream right before the call is not something a frame on another
            # lane waits for -- frames are ordered after the caller-stream work of frames_in_flight - 1 calls ago, include/pt_api.h)
            packed = torch.empty((max(r.tiles_count(rank), 1) * 1024, 4), dtype=torch.float32, device="cuda")
            r.render_tiles(packed.data_ptr())
            torch.cuda.synchronize()
            ref_full, _ = oracle_frame()
            want = tiles.pack_range(ref_full, rank, 1, 3)
            got = packed.cpu().numpy()[: want.shape[0] * 1024].reshape(want.shape)
            if not np.array_equal(got.view(np.uint32)[..., :3], want.view(np.uint32)[..., :3]):
                bad = int((got.view(np.uint32)[..., :3] != want.view(np.uint32)[..., :3]).any(-1).sum())
                stale = None
                if state.get("prev") is not None and len(state["prev"]) == len(state["spheres"]):
                    old, _ = oracle.render(state["prev"], state["materials"], state["sd"], cam, gs, threads=8, textures=state["tex"])
                    stale = bool(np.array_equal(got.view(np.uint32)[..., :3], tiles.pack_range(old, rank, 1, 3).view(np.uint32)[..., :3]))
                raise AssertionError(f"step {checked}: tiles of rank {rank}: {bad} of {got.shape[0] * 1024} pixels differ; equals the frame of the spheres before the last move: {stale}; " + " | ".join(log[-12:]))
            r.set_partition(0, 1)
            r.totals(reset=True)  # (the oracle has no per-tile ray count to compare with)
        else:                                    # whole frames, several in flight
            n = int(rng.integers(1, lanes + 1))
            bufs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
            for b in bufs:
                r.render_device(b.data_ptr())
            torch.cuda.synchronize()
            ref, ost = oracle_frame()
            expected["rays"] += n * ost.rays
            for b in bufs:  # the same frame n times: the later ones may go through the primary-beam lists
                assert np.array_equal(b.cpu().numpy().view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"step {checked}: frame; " + " | ".join(log[-12:])
        checked += 1

    try:
        with torch.cuda.stream(tstream):
            new_scene()
            r.totals(reset=True)
            for step in range(28):
                op = rng.integers(0, 15)
                if op == 0:
                    new_scene()
                elif op == 1:
                    new_textures()
                elif op == 2:
                    move(refit=True)
                elif op == 3:
                    move(refit=False)
                elif op == 4:
                    rotate()
                elif op == 5:
                    state.update(w=int(rng.choice([48, 64, 81, 160])), h=int(rng.choice([40, 48, 57, 96])))
                elif op == 6:
                    state.update(spp=int(rng.choice([1, 1, 2, 3])), bounces=int(rng.choice([0, 1, 4, 7])), di=bool(rng.random() < 0.4))
                elif op == 8 and len(state["spheres"]):
                    log.append("rebuild")
                    r.set_scene(state["spheres"], state["materials"], state["sd"])  # the moved spheres become the master scene, new topology
                    if state["tex"] is not None:
                        r.set_textures(state["tex"])
                elif op == 9:
                    rejected_call()
                elif op == 10:  # per-launch event profiling on / off, read-outs in between: no influence on any frame
                    on = bool(rng.random() < 0.5)
                    log.append(f"profiling {on}")
                    r.set_profiling(on)
                    p = r.profile(reset=bool(rng.random() < 0.5))
                    assert p.ms_traverse >= 0.0 and p.ms_tail >= 0.0
                    r.queue_sizes()
                elif op == 11:
                    log.append("synchronize")
                    r.synchronize()
                elif op == 12:  # pt_build_accel alone rebuilds the tree of the spheres of the last pt_set_scene: no effect on what the lanes hold
                    log.append("build_accel")
                    r.build_accel()
                elif op == 7:
                    state.update(pos=(float(rng.uniform(-2, 2)), float(rng.uniform(0, 2)), float(rng.uniform(-14, -9))), cam_seed=int(rng.integers(0, 64)), vel=None)
                elif op >= 13:  # start (or change) a drift: a steady translation of a few thousandths per frame, or none at all for a while
                    v = rng.normal(size=3) * float(rng.choice([0.0, 0.002, 0.004, 0.02]))
                    orbit = bool(rng.random() < 0.5)
                    log.append(f"drift {v} orbit={orbit}")
                    state.update(vel=tuple(float(x) for x in v), orbit=orbit)
                render()
            check_totals()
    finally:
        r.close()
    assert checked == 28


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_PIPE_SEEDS", "6"))))
def test_pipelined_sequences_match_oracle(dxrs, host, oracle, seed):
    """The same idea without a host synchronisation between frames: 14 frames go through the lanes back to back while the spheres move
    (with an explicit refit, without one, or not at all), object rotations change, and the view rests (beam lists are built on the side
    stream and picked up) or moves; the caller's stream snapshots every frame right after its render call.  Only then does the host wait,
    and every snapshot must be the oracle's frame -- ordering between lanes, the side stream and the caller's stream is what is on trial."""
    import torch
    from dxrs_amd import textures as T
    rng = np.random.default_rng(9000 + seed)
    lanes = (3, 2, 3, 1)[seed % 4]
    n = int(rng.choice([40, 441, 3000]))
    spheres, materials = random_scene(dxrs, rng, n)
    sd = copy.copy(host.scene(dxrs.host.SCENE_SMALL)[2])
    textured = bool(seed % 2)
    ts = None
    if textured:
        ts, env = _texture_set(dxrs, rng, n, with_env=bool(rng.random() < 0.5))
        if env is not None:
            sd.EnvironmentLightTextureDescriptor, sd.IsEnvironmentLightTextureCubeMap = env, 0
    w, h = 320, 200
    tstream = torch.cuda.Stream()
    r = dxrs.Renderer(stream=tstream.cuda_stream, frames_in_flight=lanes)
    try:
        with torch.cuda.stream(tstream):
            r.set_scene(spheres, materials, sd)
            if ts is not None:
                r.set_textures(ts)
            bufs = [torch.empty((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(lanes)]
            torch.cuda.synchronize()
            pos, cam_seed, vel, orbit = (0.0, 0.5, -12.0), 0, None, False
            snaps, frames = [], []
            for k in range(14):
                what = int(rng.integers(0, 8))
                if what == 0 or what == 1:
                    s2 = spheres.copy()
                    mv = rng.random(n) < 0.5
                    s2["cy"][mv] += rng.uniform(-0.2, 0.2, int(mv.sum())).astype(np.float32)
                    spheres = s2
                    r.update_spheres(spheres, refit=(what == 0))
                elif what == 2 and ts is not None:
                    for i in range(n):
                        if rng.random() < 0.3:
                            ts.set_rotation(i, T.quaternion_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6.28))))
                    r.update_rotations(ts.rotations)
                elif what == 3:
                    pos, cam_seed, vel = (float(rng.uniform(-1, 1)), float(rng.uniform(0, 1.5)), float(rng.uniform(-13, -10))), int(rng.integers(0, 64)), None
                elif what >= 6:  # the camera starts to drift: a steady translation without a turn (beam lists with slack, built behind a frame on its lane)
                    vel = tuple(float(x) for x in rng.normal(size=3) * float(rng.choice([0.002, 0.005])))
                    orbit = bool(rng.random() < 0.5)  # the eye stays on the origin: a slow turn on top of the travel (lists with a pixel margin, too)
                if vel is not None:
                    pos = tuple(float(np.float32(p + v)) for p, v in zip(pos, vel))
                gs = dxrs.types.graphics_settings(w, h, frame_index=k, bounces=int(rng.choice([2, 5])), spp=int(rng.choice([1, 1, 2])), di=bool(rng.random() < 0.3))
                cam = host.camera(w, h, position=pos, look_at=None if (vel is not None and not orbit) else (0.0, 0.0, 0.0), jitter_index=cam_seed)
                r.set_camera(cam); r.set_constants(gs)
                r.render_device(bufs[k % lanes].data_ptr())
                snaps.append(bufs[k % lanes].clone())  # on the caller's stream: after frame k, before the frame that reuses the buffer
                frames.append((spheres, None if ts is None else copy.deepcopy(ts.rotations), cam, gs))
            torch.cuda.synchronize()
        for k, (sph, rot, cam, gs) in enumerate(frames):
            if ts is not None:
                ts.rotations[:] = rot
            ref, _ = oracle.render(sph, materials, sd, cam, gs, threads=8, textures=ts)
            assert np.array_equal(snaps[k].cpu().numpy().view(np.uint32)[..., :3], ref.view(np.uint32)[..., :3]), f"frame {k} of sequence {seed} ({lanes} lanes, {n} spheres)"
    finally:
        r.close()
