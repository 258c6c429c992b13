"""CPU oracle for the host pieces of the bulk sampling driver.

TEST INFRASTRUCTURE - not part of the product (only ``tests/`` import it).

The reference parses its ground-truth lists inline inside ``main()`` (``full_sampling.py:132-143``, ``train.py:365-388``), so
there is no function to import; these are literal restatements of those lines, used to check ``worddiffusion_amd.driver``
on real lines of the reference's own ``gt/*`` files (``tests/golden/gt_samples.txt``: the whole of ``gt/1.txt`` plus the
first lines of ``gan.iam.test.gt copy.filter27``, ``cvlTest.txt``, ``OovAllWritersTrainSets.txt`` and
``norwegian9000_train_0_All.filter27``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple


def parse_gt_lines(lines: List[str]) -> List[Tuple[str, str, str]]:
    """full_sampling.py:132-143: ``i.strip().split(' ')``; ``s_id = i[0].split(',')[0]``; ``image = i[0].split(',')[1]``;
    ``transcription = i[1]``."""
    out = []
    for i in [ln.strip().split(' ') for ln in lines]:
        s_id = i[0].split(',')[0]
        image = i[0].split(',')[1]
        transcription = i[1]
        out.append((s_id, image, transcription))
    return out


def writer_dict_train(rows: List[Tuple[str, str, str]]) -> Dict[str, int]:
    """train.py:370-388: writers numbered in order of first appearance."""
    wr_dict, wr_index = {}, 0
    for s_id, _, _ in rows:
        if s_id not in wr_dict.keys():
            wr_dict[s_id] = wr_index
            wr_index += 1
    return wr_dict
