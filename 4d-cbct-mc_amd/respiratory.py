"""Breathing curve of a 4-D scan: which respiratory state each projection is simulated in.

Host-side mirror of the reference's `RespiratorySignal` (cbctmc/mc/respiratory.py:14-130) as far as the 4-D driver needs it
(`MCSimulation4D.run_simulation`, cbctmc/mc/simulation.py:527-710): one (amplitude, rate of change) pair per projection, optionally
quantised, and the projections grouped by identical pairs -- every group is ONE geometry warp on the device followed by its projections
on the resident context.  The numerical rules are the reference's (they decide which projections share a geometry):

    resample        linear interpolation of amplitude and rate onto int(T f) equidistant instants of [0, T]
    quantise        n equal-width classes between minimum and maximum, a value mapped to the CENTRE of its class; the maximum itself
                    falls into an (n + 1)-th class above the range (numpy.digitize's right-open classes: respiratory.py:64-70)
    group           identical (amplitude, rate) pairs in lexicographic order, each with its projection indices in ascending order
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def _grid(total_seconds: float, sampling_frequency: float) -> np.ndarray:
    return np.linspace(0.0, total_seconds, int(total_seconds * sampling_frequency))


class RespiratorySignal:
    __slots__ = ("signal", "dt_signal", "sampling_frequency", "time")

    def __init__(self, signal, dt_signal=None, sampling_frequency: float = 25.0):
        self.signal = np.asarray(signal, dtype=np.float64)
        self.sampling_frequency = float(sampling_frequency)
        # rate of change: central differences on the sampling grid unless the caller measured it
        self.dt_signal = np.gradient(self.signal, 1.0 / self.sampling_frequency) if dt_signal is None else np.asarray(dt_signal, dtype=np.float64)
        self.time = np.linspace(0.0, self.total_seconds, self.signal.size)

    @property
    def total_seconds(self) -> float:
        return self.signal.size / self.sampling_frequency

    def resample(self, sampling_frequency: float) -> "RespiratorySignal":
        """The curve at another rate (the 4-D driver asks for one sample per projection: the detector's frame rate)."""
        t = _grid(self.total_seconds, sampling_frequency)
        amplitude, rate = (np.interp(t, self.time, curve) for curve in (self.signal, self.dt_signal))
        return RespiratorySignal(amplitude, rate, sampling_frequency)

    @staticmethod
    def quantize_signal(signal, n_bins: int = 20) -> np.ndarray:
        """Class centres of `signal` for n_bins equal-width classes spanning [min, max] (module docstring: the maximum gets a class of its own)."""
        values = np.asarray(signal, dtype=np.float64)
        edges = np.linspace(values.min(), values.max(), n_bins + 1)
        width = edges[1] - edges[0]
        klass = np.searchsorted(edges, values, side="right") - 1  # edges[k] <= value < edges[k + 1]; value == max -> k = n_bins
        return edges[klass] + 0.5 * width

    @staticmethod
    def get_unique_signals(signal, dt_signal) -> Dict[Tuple[float, float], List[int]]:
        """{(amplitude, rate): [projection indices]}: one entry per respiratory state, states in lexicographic order."""
        pairs = np.column_stack((np.asarray(signal, dtype=np.float64), np.asarray(dt_signal, dtype=np.float64)))
        order = np.lexsort((pairs[:, 1], pairs[:, 0]))  # stable: indices of equal pairs stay ascending
        ranked = pairs[order]
        starts = np.flatnonzero(np.r_[True, np.any(ranked[1:] != ranked[:-1], axis=1)])
        groups = {}
        for begin, end in zip(starts, np.r_[starts[1:], len(order)]):
            groups[(float(ranked[begin, 0]), float(ranked[begin, 1]))] = order[begin:end].tolist()
        return groups

    @classmethod
    def _fourth_power(cls, wave, total_seconds, period, amplitude, sampling_frequency):
        t = _grid(total_seconds, sampling_frequency)
        # the phase in the reference's own operation order (2 pi (1 / (2 period)) t): the last bit of a sample decides its class at an edge
        return cls(amplitude * wave(2 * np.pi * (1 / (2 * period)) * t) ** 4, sampling_frequency=sampling_frequency)

    @classmethod
    def create_sin4(cls, total_seconds: float, period: float = 5.0, amplitude: float = 1.0, sampling_frequency: float = 25.0):
        """amplitude sin^4(pi t / period): a breathing cycle of `period` seconds, exhale at t = 0 (respiratory.py:95-112)."""
        return cls._fourth_power(np.sin, total_seconds, period, amplitude, sampling_frequency)

    @classmethod
    def create_cos4(cls, total_seconds: float, period: float = 5.0, amplitude: float = 1.0, sampling_frequency: float = 25.0):
        """amplitude cos^4(pi t / period): inhale at t = 0 (respiratory.py:114-130)."""
        return cls._fourth_power(np.cos, total_seconds, period, amplitude, sampling_frequency)
