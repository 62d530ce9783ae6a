"""Row N3 -- display transform (App::Impl::ToneMap -> DirectXTK ToneMapPostProcess, Source/App.cpp:1731-1757) and
progressive accumulation.  CPU: the oracle against hand-derived known answers, and the product's device header
(csrc/pt_post.h compiled as host C++ by tests/hostshim) against the oracle bit for bit.  GPU: pt_tonemap / pt_accumulate
through the C-ABI against the oracle, bit-exact (integer outputs; fp32 running mean)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.binding import declare_leaf_api

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    lib = C.CDLL(os.path.join(HERE, "hostshim", "libdevmath_host.so"))
    declare_leaf_api(lib, "dev_")
    return lib


def all_params(dxrs):
    t = dxrs.types
    out = []
    for op in (t.TONE_NONE, t.TONE_SATURATE, t.TONE_REINHARD, t.TONE_ACES_FILMIC):
        for tf in (t.TRANSFER_LINEAR, t.TRANSFER_SRGB):
            for stops in (0.0, -2.0, 1.5):
                out.append(t.tonemap_params(op, tf, stops))
    for rot in (t.ROTATE_709_TO_2020, t.ROTATE_P3D65_TO_2020, t.ROTATE_709_TO_P3D65):
        for nits in (200.0, 80.0, 10000.0):
            out.append(t.tonemap_params(t.TONE_NONE, t.TRANSFER_ST2084, 0.0, nits, rot))
    return out


def hdr_samples(rng, n):
    """radiance-like values plus everything a display transform must survive: 0, denormals, huge, inf, NaN, negatives"""
    x = np.exp(rng.uniform(np.log(1e-6), np.log(1e4), (n, 4))).astype(np.float32)
    x[rng.random((n, 4)) < 0.05] = 0.0
    special = np.array([0.0, -0.0, 1e-45, 1e-38, 1.0, 0.18, 3.4e38, np.inf, -np.inf, np.nan, -1.0, -0.5, 65504.0], dtype=np.float32)
    k = min(n // 2, 4096)
    x[:k, :3] = special[rng.integers(0, len(special), (k, 3))]
    return x


def unpack8(v):
    return np.stack([(v >> s) & 0xFF for s in (0, 8, 16, 24)], -1)


def test_oracle_known_answers(dxrs, oracle):
    t = dxrs.types
    hdr = np.array([[0.0, 0.18, 1.0, 1.0], [0.5, 2.0, 1e9, 1.0], [np.nan, -1.0, np.inf, 1.0]], dtype=np.float32)
    # Saturate + linear: round(saturate(x) * 255)
    got = unpack8(oracle.tonemap(hdr, t.tonemap_params(t.TONE_SATURATE, t.TRANSFER_LINEAR)))
    assert got.tolist() == [[0, 46, 255, 255], [128, 255, 255, 255], [0, 0, 255, 255]]
    # Reinhard x / (1 + x), then pow(., 1/2.2): 0.18 -> 0.15254 -> 0.4254 -> 108; 1 -> 0.5 -> 0.7297 -> 186
    got = unpack8(oracle.tonemap(hdr, t.tonemap_params(t.TONE_REINHARD, t.TRANSFER_SRGB)))
    ref = lambda x: int(np.floor(np.clip(x / (1 + x), 0, 1) ** (1 / 2.2) * 255 + 0.5))
    assert got[0].tolist() == [0, ref(0.18), ref(1.0), 255] and got[1, :2].tolist() == [ref(0.5), ref(2.0)]
    assert got[2, 0] == 0 and got[2, 2] == 0  # NaN and inf/(1+inf) = NaN both convert to 0
    # ACES filmic (Narkowicz): f(0.18) = 0.2670, f(1) = 0.8038; exposure +1 stop doubles the input
    aces = lambda x: np.clip(x * (2.51 * x + 0.03) / (x * (2.43 * x + 0.59) + 0.14), 0, 1)
    got = unpack8(oracle.tonemap(hdr, t.tonemap_params(t.TONE_ACES_FILMIC, t.TRANSFER_SRGB, exposure_stops=1.0)))
    exp = [int(np.floor(aces(2 * x) ** (1 / 2.2) * 255 + 0.5)) for x in (0.18, 1.0)]
    assert abs(int(got[0, 1]) - exp[0]) <= 1 and abs(int(got[0, 2]) - exp[1]) <= 1
    # ST 2084: 100 nits (paper white 100, signal 1.0) encodes to ~0.5081 -> 520/1023; grey stays grey under the rotation
    v = oracle.tonemap(np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32), t.tonemap_params(t.TONE_NONE, t.TRANSFER_ST2084, paper_white_nits=100.0))[0]
    r, g, b, a = v & 1023, (v >> 10) & 1023, (v >> 20) & 1023, v >> 30
    assert a == 3 and abs(int(r) - 520) <= 1 and abs(int(g) - 520) <= 1 and abs(int(b) - 520) <= 1
    v = oracle.tonemap(np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32), t.tonemap_params(t.TONE_NONE, t.TRANSFER_ST2084, paper_white_nits=10000.0))[0]
    assert (v & 1023) == 1023  # 10000 nits = full scale


def test_device_header_matches_oracle(dxrs, oracle, dev):
    rng = np.random.default_rng(7)
    hdr = hdr_samples(rng, 6000)
    for p in all_params(dxrs):
        want = oracle.tonemap(hdr, p)
        got = np.array([dev.dev_tonemap_pixel(hdr[i].ctypes.data_as(C.POINTER(C.c_float)), C.addressof(p)) for i in range(0, len(hdr), 3)], dtype=np.uint32)
        assert np.array_equal(got, want[::3]), (p.Operator, p.TransferFunction, p.ColorRotation)
    # running mean
    a = np.zeros((500, 4), dtype=np.float32); b = a.copy()
    for n in range(6):
        x = hdr_samples(rng, 500)
        x[~np.isfinite(x)] = 1.0
        oracle.accumulate(a, x, n)
        dev.dev_accumulate(b.ctypes.data, x.ctypes.data, 500, n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_accumulate_is_the_running_mean(oracle):
    rng = np.random.default_rng(3)
    frames = rng.random((32, 100, 4)).astype(np.float32)
    acc = np.zeros((100, 4), dtype=np.float32)
    for n, f in enumerate(frames):
        oracle.accumulate(acc, f, n)
    assert np.allclose(acc, frames.astype(np.float64).mean(0), rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_gpu_tonemap_and_accumulate_match_oracle(dxrs, host, oracle, renderer):
    import torch
    rng = np.random.default_rng(11)
    # a rendered frame (real radiance distribution) + adversarial values
    spheres, materials, sd = host.scene(dxrs.host.SCENE_DEMO, seed=0)
    w, h = 320, 180
    renderer.set_scene(spheres, materials, sd); renderer.set_camera(host.camera(w, h)); renderer.set_constants(dxrs.types.graphics_settings(w, h, bounces=4))
    img, _ = renderer.render()
    hdr = np.concatenate([img.reshape(-1, 4), hdr_samples(rng, 20000)]).astype(np.float32)
    d_hdr = torch.from_numpy(hdr).cuda()
    d_out = torch.zeros(len(hdr), dtype=torch.int32, device="cuda")
    for p in all_params(dxrs):
        renderer.tonemap(d_hdr.data_ptr(), len(hdr), p, d_out.data_ptr())
        renderer.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.tonemap(hdr, p)), (p.Operator, p.TransferFunction, p.ColorRotation)
    # progressive accumulation of 8 jittered frames == oracle running mean, bit for bit; noise goes down
    acc_ref = np.zeros((h, w, 4), dtype=np.float32)
    d_acc = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    d_frame = torch.zeros((h * w, 4), dtype=torch.float32, device="cuda")
    gs = dxrs.types.graphics_settings(w, h, bounces=4)
    first = None
    for n in range(8):
        gs.FrameIndex = n
        renderer.set_camera(host.camera(w, h, jitter_index=n)); renderer.set_constants(gs)
        renderer.render_device(d_frame.data_ptr())
        renderer.accumulate(d_acc.data_ptr(), d_frame.data_ptr(), h * w, n)
        renderer.synchronize()
        f = d_frame.cpu().numpy().reshape(h, w, 4)
        first = f.copy() if first is None else first
        oracle.accumulate(acc_ref, f, n)
        assert np.array_equal(d_acc.cpu().numpy().reshape(h, w, 4).view(np.uint32), acc_ref.view(np.uint32))
    rough = lambda a: np.abs(np.diff(a[..., :3], axis=1)).mean()  # edges stay, Monte-Carlo noise averages out
    assert rough(acc_ref) < 0.9 * rough(first)
    with pytest.raises(RuntimeError):
        bad = dxrs.types.tonemap_params(); bad.Operator = 9
        renderer.tonemap(d_hdr.data_ptr(), 4, bad, d_out.data_ptr())
