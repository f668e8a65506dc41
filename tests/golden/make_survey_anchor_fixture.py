#!/usr/bin/env python3
"""Build tests/golden/survey_anchors_C1.npz: the ONE fixture in this repo that is pinned by something outside this repo's code.

PROVENANCE.  SURVEY.md Appendix C records eight anchor pixels (premultiplied R and A, to 5 decimals) of config C1 rendered by the
surveyor's own throwaway float64 numpy probe, written from the reference's shader text independently of this repo's oracle, with
the reference's displacement cubemap asset as input ("reference cubemap R-channel/255 with the B.5 face convention").  This script
stores
  * cubemap_r : uint8 [6,128,128], the R bytes (ARGB32: byte 1 of each texel) of the `_typelessdata` blob of
                /root/reference/Assets/Textures/DisplacementTexture.cubemap, faces and rows in the asset's order -- DATA of the
                reference (the texture the demo binds as _DisplacementTexture, FillVolume.mat:23-29), not source;
  * cols, rows, r, a : the eight anchors exactly as printed in SURVEY.md App. C.
The expected values come from the survey, NOT from this repo's oracle: tests/test_oracle_golden.py checks the oracle (and
tests/test_gpu_parity.py the HIP path) against them.  Needs /root/reference (this container only).

Run from the repo root:  python tests/golden/make_survey_anchor_fixture.py
"""
import os
import re

import numpy as np

ASSET = "/root/reference/Assets/Textures/DisplacementTexture.cubemap"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "survey_anchors_C1.npz")

# (col, row) -> (premultiplied R, A); row 0 = bottom of the view.  SURVEY.md Appendix C.
ANCHORS = {(182, 55): (0.45312, 0.99968), (71, 81): (0.45175, 0.99393), (71, 181): (0.46475, 0.99839), (192, 142): (0.46979, 1.00000),
           (46, 56): (0.46420, 0.98732), (98, 116): (0.47258, 1.00000), (149, 124): (0.41245, 0.99999), (86, 68): (0.44885, 0.98690)}

if __name__ == "__main__":
    txt = open(ASSET).read()
    assert "m_Width: 128" in txt and "m_TextureFormat: 5" in txt and "m_ImageCount: 6" in txt
    blob = bytes.fromhex(re.search(r"_typelessdata:\s*([0-9a-f]+)", txt).group(1))
    raw = np.frombuffer(blob, dtype=np.uint8).reshape(6, 128, 128, 4)          # ARGB32
    np.savez_compressed(OUT, cubemap_r=np.ascontiguousarray(raw[..., 1]),
                        cols=np.array([c for c, _ in ANCHORS], dtype=np.int32), rows=np.array([r for _, r in ANCHORS], dtype=np.int32),
                        r=np.array([v[0] for v in ANCHORS.values()], dtype=np.float64), a=np.array([v[1] for v in ANCHORS.values()], dtype=np.float64))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
