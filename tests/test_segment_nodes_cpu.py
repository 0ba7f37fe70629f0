"""SegmentSplit / SegmentMerge (ttt_amd/models/cogvideo/dit.py): the multi-segment local attention's glue as two autograd nodes.
They must reproduce - bit for bit, forward and backward, in fp32 and in bf16 - the statement-by-statement form the reference
uses (slice + cat per segment, accumulate into zeros, count, divide, cat: /root/reference/ttt/models/cogvideo/dit.py:163-211),
restated below as the checker."""
import pytest
import torch

from ttt_amd.models.cogvideo.dit import SegmentMerge, SegmentSplit, _segment_geometry
from ttt_amd.models.cogvideo.utils import SequenceMetadata


def statement_form(vid_emb, text_emb, meta, attn_length, prefix, segment):
    tl, tpf = meta.text_length, meta.tokens_per_frame
    out_vid = torch.zeros_like(vid_emb)
    out_txt = torch.zeros_like(text_emb)
    count = torch.zeros_like(vid_emb[..., :1])
    for i in range(meta.num_chunks):
        lo = i * attn_length * tpf
        hi = (prefix + (i + 1) * attn_length) * tpf
        seg = torch.cat((text_emb[:, i * tl:(i + 1) * tl], vid_emb[:, lo:hi]), dim=1)
        o = segment(seg, i)
        out_txt[:, i * tl:(i + 1) * tl] = o[:, :tl]
        out_vid[:, lo:hi] += o[:, tl:]
        count[:, lo:hi] += 1
    return torch.cat((out_txt, out_vid / count), dim=1)


def node_form(vid_emb, text_emb, meta, attn_length, prefix, segment):
    x = torch.cat((text_emb, vid_emb), dim=1)
    n_text = text_emb.shape[1]
    tl, rng, shared = _segment_geometry(meta, vid_emb.shape[1], attn_length, prefix)
    segs = SegmentSplit.apply(x, n_text, tl, rng, shared)
    return SegmentMerge.apply(n_text, tl, rng, shared, *(segment(s, i) for i, s in enumerate(segs)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geo", [(2, 3, 4, 1, 5, 7, 0), (1, 4, 3, 1, 8, 5, 0), (2, 2, 5, 2, 3, 4, 0), (1, 3, 4, 1, 6, 3, 2)])
def test_segment_nodes_equal_statement_form_bitwise(geo, dtype):
    B, n_seg, attn_length, prefix, tpf, tl, cut = geo            # cut: frames the video ends short of the last segment
    D = 16
    frames = prefix + n_seg * attn_length - cut
    meta = SequenceMetadata(text_length=tl, seq_text_length=n_seg * tl, num_frames=frames, num_chunks=n_seg, tokens_per_frame=tpf,
                            latent_height=1, latent_width=tpf, t_emb=torch.zeros(1, 4))
    g = torch.Generator().manual_seed(sum(geo))
    vid0 = torch.randn(B, frames * tpf, D, generator=g).to(dtype)
    txt0 = torch.randn(B, n_seg * tl, D, generator=g).to(dtype)
    ws = [torch.randn(D, D, generator=g).to(dtype) for _ in range(n_seg)]
    dy = torch.randn(B, n_seg * tl + frames * tpf, D, generator=g).to(dtype)
    segment = lambda s, i: torch.tanh(s @ ws[i]) * (1.0 + 0.1 * i)           # any per-segment map that mixes the rows' features

    res = []
    for form in (statement_form, node_form):
        vid, txt = vid0.clone().requires_grad_(True), txt0.clone().requires_grad_(True)
        y = form(vid, txt, meta, attn_length, prefix, segment)
        y.backward(dy)
        res.append((y.detach(), vid.grad, txt.grad))
    for a, b, name in zip(res[0], res[1], ("out", "d_video", "d_text")):
        assert a.shape == b.shape and torch.equal(a, b), (name, (a.float() - b.float()).abs().max())


def test_segment_geometry_rejects_layouts_that_do_not_tile_the_video():
    meta = SequenceMetadata(text_length=2, seq_text_length=4, num_frames=9, num_chunks=2, tokens_per_frame=3, latent_height=1,
                            latent_width=3, t_emb=torch.zeros(1, 4))
    _segment_geometry(meta, 9 * 3, 4, 1)
    _segment_geometry(meta, 8 * 3, 4, 1)              # last segment cut short by the end of the video: as the reference's slices
    with pytest.raises(AssertionError):
        _segment_geometry(meta, 10 * 3, 4, 1)         # a frame no segment covers
    with pytest.raises(AssertionError):
        _segment_geometry(meta, 5 * 3, 4, 1)          # second segment = its shared frame only
