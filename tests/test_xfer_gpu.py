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
