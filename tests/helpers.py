"""Shared helpers for the tests (no reference access at run time)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pattern_pairs():
    """The 256 (x0,y0,x1,y1) rBRIEF test pairs from include/slamit_orb_pattern.h."""
    txt = open(os.path.join(ROOT, "include", "slamit_orb_pattern.h")).read()
    body = txt[txt.index("= {") + 3: txt.rindex("};")]
    nums = [int(v) for v in re.findall(r"-?\d+", body)]
    assert len(nums) == 1024
    return np.array(nums, np.int32).reshape(256, 4)
