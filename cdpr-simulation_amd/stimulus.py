"""Host-side counterparts of the reference's stimulus publishers (src/*test.cpp).

Each generator yields the float32 `Joy.axes` sequence its ROS node would publish,
one sample per publisher tick.  The reference nodes run on wall-clock `ros::Rate`
and accumulate `time += 1.0 / rate` in double (sinevelocitytest.cpp:35,48), so the
accumulation is reproduced, not replaced by k / rate.
"""
from __future__ import annotations

import math
from typing import Iterator

import numpy as np


def _ticks(rate_hz: float) -> Iterator[float]:
    t = 0.0
    while True:
        yield t
        t += 1.0 / rate_hz


def sine_velocity(n_cables: int = 4, amp: float = 0.05, freq: float = 0.1, rate_hz: float = 100.0) -> Iterator[np.ndarray]:
    """sinevelocitytest.cpp:6-10,34-49: v = amp * sin(t * freq * 2 * pi) on every axis, 100 Hz."""
    for t in _ticks(rate_hz):
        v = amp * math.sin(t * freq * 2 * math.pi)
        yield np.full(n_cables, v, dtype=np.float32)


def square_velocity(n_cables: int = 4, amp: float = 0.06, freq: float = 0.05, rate_hz: float = 10.0) -> Iterator[np.ndarray]:
    """squarevelocitytest.cpp:6-9,20-34: +-amp while |sin| >= sqrt(1/2), else 0, 10 Hz.
    The reference writes unqualified abs(sine) (line 22, int-abs hazard); fabs here."""
    for t in _ticks(rate_hz):
        s = math.sin(t * freq * 2 * math.pi)
        v = math.copysign(amp, s) if abs(s) >= math.sqrt(0.5) else 0.0
        yield np.full(n_cables, v, dtype=np.float32)


def square_position(n_cables: int = 4, amp: float = 0.05, bias: float = 0.0, freq: float = 0.1, rate_hz: float = 10.0) -> Iterator[np.ndarray]:
    """squarepositiontest.cpp:6-10,21-35: bias + copysign(amp, sin), 10 Hz."""
    for t in _ticks(rate_hz):
        s = math.sin(t * freq * 2 * math.pi)
        yield np.full(n_cables, bias + math.copysign(amp, s), dtype=np.float32)
