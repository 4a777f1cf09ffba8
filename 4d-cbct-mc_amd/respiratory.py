"""`RespiratorySignal` host-side mirror (cbctmc/mc/respiratory.py:14-130): the breathing curve that selects, per projection,
which warped geometry is simulated.  Same resampling, quantisation and grouping rules as the reference."""
from __future__ import annotations

import numpy as np


class RespiratorySignal:
    def __init__(self, signal, dt_signal=None, sampling_frequency: float = 25.0):
        self.signal = np.asarray(signal, dtype=np.float64)
        self.sampling_frequency = sampling_frequency
        self.dt_signal = np.asarray(dt_signal, dtype=np.float64) if dt_signal is not None else np.gradient(self.signal, 1 / sampling_frequency)
        self.time = np.linspace(0, self.total_seconds, len(self.signal))

    @property
    def total_seconds(self):
        return len(self.signal) / self.sampling_frequency

    def resample(self, sampling_frequency: float) -> "RespiratorySignal":
        """respiratory.py:45-55: linear interpolation onto int(T * f) samples."""
        t = np.linspace(0, self.total_seconds, int(self.total_seconds * sampling_frequency))
        return RespiratorySignal(np.interp(t, self.time, self.signal), np.interp(t, self.time, self.dt_signal), sampling_frequency)

    @staticmethod
    def quantize_signal(signal, n_bins: int = 20):
        """respiratory.py:64-70: bin centres of n_bins equal-width bins between min and max."""
        signal = np.asarray(signal)
        bins = np.linspace(signal.min(), signal.max(), n_bins + 1)
        idx = np.digitize(signal, bins=bins)
        return bins[idx - 1] + 0.5 * (bins[1] - bins[0])

    @staticmethod
    def get_unique_signals(signal, dt_signal):
        """respiratory.py:79-93: {(signal, dt_signal): [projection indices]} in np.unique order."""
        samples = np.stack((signal, dt_signal), axis=-1)
        out = {}
        for u in np.unique(samples, axis=0):
            out[tuple(u.tolist())] = np.where((samples == u).all(axis=1))[0].tolist()
        return out

    @classmethod
    def create_sin4(cls, total_seconds: float, period: float = 5.0, amplitude: float = 1.0, sampling_frequency: float = 25.0):
        t = np.linspace(0, total_seconds, int(total_seconds * sampling_frequency))
        return cls(amplitude * np.sin(2 * np.pi * (1 / (2 * period)) * t) ** 4, sampling_frequency=sampling_frequency)

    @classmethod
    def create_cos4(cls, total_seconds: float, period: float = 5.0, amplitude: float = 1.0, sampling_frequency: float = 25.0):
        t = np.linspace(0, total_seconds, int(total_seconds * sampling_frequency))
        return cls(amplitude * np.cos(2 * np.pi * (1 / (2 * period)) * t) ** 4, sampling_frequency=sampling_frequency)
