"""The node boundary's coded transfers (vstab_upload_f32_coded / vstab_download_mask_coded, include/vstab.h): fewer bytes on
PCIe where the data allow it, the destination's BITS equal to the source's either way."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
QCHUNK = 32 << 20                 # values per coded chunk (vstab_xfer.hip)


def bits(t):
    import torch

    return t.contiguous().view(torch.int32)


def eight_bit_values(n, seed):
    """What a ComfyUI IMAGE decoded from 8-bit video holds: float32(k) / float32(255) (nodes/stabilizer_utils.py:122-126)."""
    import torch

    g = torch.Generator().manual_seed(seed)
    k = torch.randint(0, 256, (n,), generator=g, dtype=torch.int32)
    return torch.from_numpy(k.numpy().astype(np.float32) / np.float32(255.0))


@pytest.mark.parametrize("n", [1, 7, 1000, (1 << 20) + 3, QCHUNK, QCHUNK * 2 + 12, QCHUNK * 5 - 8])
def test_coded_upload_of_8bit_sourced_values_is_bit_exact(ctx, n):
    """Sizes below the threading threshold, one chunk exactly, several turns of the ring, ragged tails (n % 16 != 0): every
    chunk crosses as bytes and the device tensor has the source's bits; a torch op queued right behind the call sees them."""
    import torch

    host = eight_bit_values(n, n % 977)
    dev = ctx.upload(host)
    nxt = dev * 1.0                                   # stream-ordered consumer, no synchronisation in between
    chunks = -(-n // QCHUNK)
    assert ctx.last_upload_coded == (chunks, chunks)
    assert torch.equal(bits(dev.cpu()), bits(host)) and torch.equal(bits(nxt.cpu()), bits(host))
    again = ctx.upload(host[: max(1, n // 3)])        # the ring and the device slots are reused by the next call
    assert torch.equal(bits(again.cpu()), bits(host[: max(1, n // 3)]))


@pytest.mark.parametrize("special", [0.5, float("nan"), -0.0, 1.0000001, -1e-9, 255.0, float("inf"), 1.0 / 255.0 * 0.9999999])
def test_coded_upload_keeps_the_bits_of_values_that_are_not_quotients(ctx, special):
    """One value that is not float32(k) / 255 in the SECOND chunk: chunk 0 crosses as bytes, chunk 1 and everything behind it as
    float32; the same value in chunk 0: nothing is coded.  Bits preserved in every case
    (-0.0 and NaN included: the test compares integer views)."""
    import torch

    host = eight_bit_values(QCHUNK * 2 + 100, 5)
    host[QCHUNK + 12345] = special
    dev = ctx.upload(host)
    assert ctx.last_upload_coded == (1, 3)
    assert torch.equal(bits(dev.cpu()), bits(host))
    host[17] = special
    dev = ctx.upload(host)
    assert ctx.last_upload_coded == (0, 3)
    assert torch.equal(bits(dev.cpu()), bits(host))


def test_coded_upload_of_ordinary_floats_and_the_switch(ctx, monkeypatch):
    import torch

    host = torch.rand((QCHUNK + 5,), generator=torch.Generator().manual_seed(1))
    dev = ctx.upload(host)
    assert ctx.last_upload_coded == (0, 2) and torch.equal(bits(dev.cpu()), bits(host))
    monkeypatch.setenv("VSTAB_XFER_CODED", "0")
    q = eight_bit_values(QCHUNK + 5, 2)
    dev = ctx.upload(q)
    assert ctx.last_upload_coded == (0, 2) and torch.equal(bits(dev.cpu()), bits(q))


@pytest.mark.parametrize("n", [5, (1 << 20) - 1, (1 << 20) + 5, (32 << 20) + 17, (32 << 20) * 5 - 3])
def test_coded_mask_download(ctx, n):
    """A mask of zeros and ones crosses as bytes (from 2^20 values on; below that the plain path) and comes back with its bits;
    a download queued right behind the kernel that produced the mask sees its result."""
    import torch

    g = torch.Generator().manual_seed(n % 991)
    host = (torch.rand((n,), generator=g) < 0.3).to(torch.float32)
    dev = host.cuda()
    back = ctx.download(dev, mask=True)
    assert ctx.last_download_coded == (n >= (1 << 20))
    assert back.device.type == "cpu" and torch.equal(bits(back), bits(host))
    flipped = 1.0 - dev                               # stream-ordered producer
    assert torch.equal(bits(ctx.download(flipped, mask=True)), bits(1.0 - host))


@pytest.mark.parametrize("special", [0.5, -0.0, float("nan"), 1.0 / 33.0, 2.0])
def test_soft_masks_take_the_plain_download(ctx, special):
    """Motion Apply's mask under motion blur is 1 - coverage / S (nodes/motion_apply.py:195-199): any value that is not 0.0f or
    1.0f by its bits sends the whole mask through vstab_download."""
    import torch

    n = (1 << 21) + 9
    host = (torch.rand((n,), generator=torch.Generator().manual_seed(3)) < 0.5).to(torch.float32)
    for pos in (0, n // 2, n - 1):
        m = host.clone()
        m[pos] = special
        back = ctx.download(m.cuda(), mask=True)
        assert ctx.last_download_coded is False and torch.equal(bits(back), bits(m))


def test_flow_node_outputs_do_not_depend_on_the_coded_transfers(pkg, ctx, monkeypatch):
    """The Flow node on an 8-bit-sourced CPU clip, CPU tensors back: frames, masks and meta identical with the coded transfers
    on (the clip crosses as bytes, the mask comes back as bytes) and off."""
    import torch

    from tests.util import synth_frames
    from vstab_amd import nodes

    frames = synth_frames(6, 540, 960, seed=4)
    frames = (np.clip(np.round(frames * 255.0), 0, 255).astype(np.float32) / np.float32(255.0))
    args = (16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    out = nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), *args)
    assert ctx.last_upload_coded[0] == ctx.last_upload_coded[1] >= 1 and ctx.last_download_coded is True
    monkeypatch.setenv("VSTAB_XFER_CODED", "0")
    ref = nodes.VideoStabilizerFlow.execute(torch.from_numpy(frames), *args)
    assert ctx.last_upload_coded[0] == 0 and ctx.last_download_coded is False
    assert torch.equal(bits(out[0]), bits(ref[0])) and torch.equal(bits(out[1]), bits(ref[1])) and out[2] == ref[2]


@pytest.mark.parametrize("n", [1, 7, 1000, (1 << 20) + 3, QCHUNK + 5, QCHUNK * 5 - 8])
def test_uint8_upload_is_numpys_division(ctx, n):
    """vstab_upload_u8_as_f32: the device tensor has the bits of `arr.astype(np.float32) / 255.0` (stabilizer_utils.py:122-126)."""
    import torch

    k = torch.randint(0, 256, (n,), generator=torch.Generator().manual_seed(n % 983), dtype=torch.uint8)
    ref = k.numpy().astype(np.float32)
    ref /= 255.0
    dev = ctx.upload_u8_as_f32(k)
    assert dev.dtype == torch.float32 and torch.equal(bits(dev.cpu()), bits(torch.from_numpy(ref)))


def test_uint8_clips_through_the_nodes_equal_their_float_form(pkg, ctx):
    """F0's uint8 branch at the node boundary: a uint8 [N,H,W,3] CPU tensor gives what the same clip as float32(k) / 255 gives --
    frames, masks, meta -- through the Flow node (normal path, single frame, crop bypass) and Motion Apply; the adapter records
    the uint8 origin as the reference's does."""
    import torch

    from tests.util import synth_frames
    from vstab_amd import host_math as hm
    from vstab_amd import nodes

    u8 = np.clip(np.round(synth_frames(6, 270, 480, seed=9) * 255.0), 0, 255).astype(np.uint8)
    f32 = u8.astype(np.float32)
    f32 /= 255.0
    c = hm._normalize_video_input(torch.from_numpy(u8))
    assert c.adapter.dtype == np.uint8 and c.adapter.value_range == "0_255" and not c.range_pending and (c.width, c.height) == (480, 270)
    slow = hm._normalize_video_input([torch.from_numpy(f) for f in u8])          # the per-frame path of the reference
    assert (slow.adapter.dtype, slow.adapter.value_range, slow.adapter.origin) == (c.adapter.dtype, c.adapter.value_range, c.adapter.origin)
    assert torch.equal(bits(c.device_batch(ctx).cpu()), bits(torch.from_numpy(f32)))
    for args in [(16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F"),
                 (16.0, "crop", "similarity", False, 0.7, 0.5, 1.0, "#102030"),              # crop bypass: the original frames come back
                 (16.0, "expand", "translation", False, 0.7, 0.5, 0.6, "#7F7F7F")]:
        a = nodes.VideoStabilizerFlow.execute(torch.from_numpy(u8), *args)
        b = nodes.VideoStabilizerFlow.execute(torch.from_numpy(f32), *args)
        assert torch.equal(bits(a[0]), bits(b[0])) and torch.equal(bits(a[1]), bits(b[1])) and a[2] == b[2], args
    a = nodes.VideoStabilizerFlow.execute(torch.from_numpy(u8[:1]), 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")
    assert torch.equal(bits(a[0]), bits(torch.from_numpy(f32[:1])))                 # single-frame passthrough
    meta = nodes.VideoStabilizerFlow.execute(torch.from_numpy(f32), 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")[2]
    for blur, quality in ((0.0, "Standard"), (0.5, "High")):
        a = nodes.VideoStabilizerMotionApply.execute(torch.from_numpy(u8), meta, "crop_and_pad", "bilinear", "#7F7F7F", blur, quality)
        b = nodes.VideoStabilizerMotionApply.execute(torch.from_numpy(f32), meta, "crop_and_pad", "bilinear", "#7F7F7F", blur, quality)
        assert torch.equal(bits(a[0]), bits(b[0])) and torch.equal(bits(a[1]), bits(b[1]))


def soft_mask(n, levels, seed):
    """What warp_blur_kernel writes: 1 - c / S in float32 for c = 0 .. S covered samples, values below 1e-3 set to 0."""
    import torch

    c = torch.randint(0, levels + 1, (n,), generator=torch.Generator().manual_seed(seed), dtype=torch.int32).numpy()
    m = np.float32(1.0) - c.astype(np.float32) / np.float32(levels)
    m[m < np.float32(1e-3)] = 0.0
    return torch.from_numpy(m)


@pytest.mark.parametrize("levels", [1, 3, 5, 9, 17, 33])
@pytest.mark.parametrize("n", [(1 << 20) + 5, (32 << 20) + 17])
def test_soft_mask_levels_cross_as_bytes(ctx, levels, n):
    """vstab_download_mask_levels: the S + 1 values of a motion-blurred mask come back with their bits; the same mask declared with
    another S, or with one value off by an ulp, takes the plain path (and still comes back with its bits)."""
    import torch

    host = soft_mask(n, levels, levels)
    back = ctx.download(host.cuda(), mask=True, levels=levels)
    assert ctx.last_download_coded is True and torch.equal(bits(back), bits(host))
    if levels > 1:
        other = ctx.download(host.cuda(), mask=True, levels=levels - 1 if levels != 3 else 7)
        assert ctx.last_download_coded is False and torch.equal(bits(other), bits(host))
        m = host.clone()
        pos = int(torch.nonzero((m > 0) & (m < 1))[0])
        m[pos] = torch.nextafter(m[pos], torch.tensor(2.0))
        again = ctx.download(m.cuda(), mask=True, levels=levels)
        assert ctx.last_download_coded is False and torch.equal(bits(again), bits(m))


def test_motion_apply_node_sends_its_soft_mask_as_bytes(pkg, ctx, monkeypatch):
    import torch

    from tests.util import synth_frames
    from vstab_amd import nodes

    frames = torch.from_numpy(synth_frames(6, 540, 960, seed=8))
    meta = nodes.VideoStabilizerFlow.execute(frames, 16.0, "crop_and_pad", "similarity", False, 0.7, 0.5, 0.6, "#7F7F7F")[2]
    out = nodes.VideoStabilizerMotionApply.execute(frames, meta, "crop_and_pad", "bicubic", "#7F7F7F", 0.5, "High")
    assert ctx.last_download_coded is True
    soft = out[1]
    assert float(soft.max()) <= 1.0 and bool(((soft > 0) & (soft < 1)).any())          # a genuinely soft mask
    monkeypatch.setenv("VSTAB_XFER_CODED", "0")
    ref = nodes.VideoStabilizerMotionApply.execute(frames, meta, "crop_and_pad", "bicubic", "#7F7F7F", 0.5, "High")
    assert ctx.last_download_coded is False
    assert torch.equal(bits(out[0]), bits(ref[0])) and torch.equal(bits(out[1]), bits(ref[1])) and out[2] == ref[2]
