"""Error metrics of ``data/evaluation.py:9-18,63-103`` (reference)."""
from __future__ import annotations

import numpy as np


class Evaluation(object):
    @classmethod
    def maxJntError(cls, skel1, skel2):
        diff = np.asarray(skel1).reshape(-1, 3) - np.asarray(skel2).reshape(-1, 3)
        return float(np.linalg.norm(diff, axis=1).max())

    @classmethod
    def meanJntError(cls, skel1, skel2):
        diff = np.asarray(skel1).reshape(-1, 3) - np.asarray(skel2).reshape(-1, 3)
        return float(np.linalg.norm(diff, axis=1).mean())

    @classmethod
    def plotError(cls, values, path, thresholds=None):
        """Fraction of frames whose max joint error is below each threshold (the curve file of :63-103)."""
        values = np.sort(np.asarray(values, np.float64))
        thresholds = np.arange(0, 85, 1.0) if thresholds is None else np.asarray(thresholds, np.float64)
        frac = np.searchsorted(values, thresholds, side='right') / max(len(values), 1)
        with open(path, 'w') as f:
            for t, v in zip(thresholds, frac):
                f.write('%.1f\t%.6f\n' % (t, v))
        return thresholds, frac
