"""Error metrics and the error-curve file of ``data/evaluation.py`` (reference).

``maxJntError`` / ``meanJntError`` follow :9-18.  ``averageMaxJntError`` (:21-61) and ``plotError`` (:63-103)
share one curve: 17 thresholds ``5*t + 0.5`` mm (t = 0..16), the share of frames whose score is STRICTLY below the
threshold; ``plotError`` writes it as ``'%f %f\\n' % (threshold, percent)`` with the share multiplied by 100, and both
print the share (a fraction, not a percent -- the reference's wording) of frames at or below 10.5 / 20.5 / 30.5 /
40.5 mm first.  Any consumer of the reference's ``*_error.txt`` reads this file unchanged.
"""
from __future__ import annotations

import sys

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

    @staticmethod
    def _report(score_list, log):
        """The four printed lines (:66-88): share of frames with score <= 10.5 / 20.5 / 30.5 / 40.5."""
        scores = np.asarray(score_list, np.float64)
        n = max(len(scores), 1)
        if log is True:                                   # resolved at call time (the reference prints to stdout)
            log = sys.stdout
        for mm in (10, 20, 30, 40):
            share = float((scores <= mm + 0.5).sum()) / n
            if log:
                print('%dmm percentage: %f' % (mm, share), file=log)

    @staticmethod
    def _curve(score_list):
        """thresh_list, precent_list of :90-100: strict '<', fractions in [0, 1]."""
        scores = np.asarray(score_list, np.float64)
        n = max(len(scores), 1)
        thresh_list = [t * 5.0 + 0.5 for t in range(0, 17)]
        precent_list = [float((scores < th).sum()) / n for th in thresh_list]
        return thresh_list, precent_list

    @classmethod
    def averageMaxJntError(cls, score_list, log=True):
        """(:21-61) prints the four shares and returns (thresh_list, precent_list)."""
        cls._report(score_list, log)
        return cls._curve(score_list)

    @classmethod
    def plotError(cls, score_list, fig_path, log=True):
        """(:63-103) prints the four shares and writes the curve file: one '%f %f' line per threshold, percent x 100."""
        cls._report(score_list, log)
        thresh_list, precent_list = cls._curve(score_list)
        with open(fig_path, 'w') as f:
            for thresh, p in zip(thresh_list, precent_list):
                f.write('%f %f\n' % (thresh, p * 100.))
        return thresh_list, precent_list
