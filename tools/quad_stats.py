#!/usr/bin/env python3
"""Would a wave that owns FOUR neighbouring positions of a line (instead of two) save the fused aggregation launch work?
Window statistics of aligned groups of four positions from the oracle's arms (CPU only, report only):
  nice      share of groups whose four windows are nested in order (starts and ends non-decreasing along the line, a
            common core): only those could be summed as chains that switch on and off one by one without per-term masks
  leaders   distinct window starts per group (windows with a common start share their chain of additions)
  head/tail terms in front of the last start / behind the first end (each needs a per-window decision)
  adds      additions per group: today's pairs (chain shared when the two starts coincide) against chains shared
            among all windows of the group with a common start; reads: ring slots read per group, pairs against a union
"""
import sys, importlib, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import oracle as orc
synth = importlib.import_module("stereo-matching_amd.synth")
from PIL import Image
def stats(L, name):
    a = orc.build_arms(L).astype(int)   # H,W,4: left,right,top,bottom
    H, W = a.shape[:2]
    for axis, (ni, pi) in (("H", (0, 1)), ("V", (2, 3))):
        neg, pos = a[..., ni], a[..., pi]
        if axis == "V":
            neg, pos = neg.T, pos.T      # lines = columns
        LEN = neg.shape[1]; n4 = LEN // 4
        idx = np.arange(n4 * 4).reshape(n4, 4)
        b = (idx[None] - neg[:, :n4*4].reshape(-1, n4, 4))
        e = (idx[None] + pos[:, :n4*4].reshape(-1, n4, 4))
        mono = (np.diff(b, axis=2) >= 0).all(2) & (np.diff(e, axis=2) >= 0).all(2)
        core = e.min(2) - b.max(2) + 1
        nice = mono & (core >= 1)
        leaders = 1 + (np.diff(b, axis=2) > 0).sum(2)
        head = b.max(2) - b.min(2); tail = e.max(2) - e.min(2)
        union = e.max(2) - b.min(2) + 1
        wsum = (e - b + 1).sum(2)
        # adds with leader sharing (nice quads): head/core adds per leader, tails per window
        # leader chains: for each distinct start, terms from its start to min e (end of core); then tails per window
        adds_shared = np.zeros_like(wsum)
        bs = np.sort(b, axis=2)
        first = np.concatenate([np.ones(bs.shape[:2] + (1,), bool), np.diff(bs, axis=2) > 0], axis=2)
        adds_shared = ((e.min(2)[..., None] - bs + 1) * first).sum(2) + (e - e.min(2)[..., None]).sum(2)
        # pair scheme of today: pairs (0,1),(2,3): shared if equal start
        def pair_adds(j):
            b0, b1, e0, e1 = b[..., j], b[..., j+1], e[..., j], e[..., j+1]
            both = (e0 - b0 + 1) + (e1 - b1 + 1)
            sh = np.where(b0 == b1, np.minimum(e0, e1) - b0 + 1, 0)
            return both - sh
        adds_pair = pair_adds(0) + pair_adds(2)
        pair_reads = (np.maximum(e[..., 0], e[..., 1]) - np.minimum(b[..., 0], b[..., 1]) + 1) + (np.maximum(e[..., 2], e[..., 3]) - np.minimum(b[..., 2], b[..., 3]) + 1)
        print("%s %s: nice %.3f mono %.3f  leaders mean %.2f (1:%.2f 2:%.2f 3:%.2f 4:%.2f)  head %.2f tail %.2f  union %.1f  sum of windows %.1f  adds pair-scheme %.1f  quad-shared %.1f (nice only: %.1f vs %.1f)  reads pair %.1f quad %.1f"
              % (name, axis, nice.mean(), mono.mean(), leaders.mean(), *[(leaders == k).mean() for k in (1, 2, 3, 4)], head.mean(), tail.mean(), union.mean(), wsum.mean(), adds_pair.mean(), adds_shared.mean(), adds_pair[nice].mean(), adds_shared[nice].mean(), pair_reads.mean(), union.mean()))
L, R, _ = synth.pair(1000, 1920, 1080, 256)
stats(L, "synth1000")
im = np.asarray(Image.open(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/tests/data/artroom1/im0.png').convert('RGB'))[..., ::-1].copy()
stats(im, "artroom1")
im = np.asarray(Image.open(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/tests/data/teddy/im2.png').convert('RGB'))[..., ::-1].copy() if __import__('os').path.exists(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))) + '/tests/data/teddy/im2.png') else None
if im is not None: stats(im, "teddy")
