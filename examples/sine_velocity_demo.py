#!/usr/bin/env python3
"""Drives the drop-in facade exactly as the reference's `sinevelocitytest` node drives the Gazebo plugin: a 100 Hz
sine on `jointVelocities`, the platform pose read back from `platformPose` — for 1 robot or a whole batch.

    python examples/sine_velocity_demo.py [robots] [seconds]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cdpr_simulation_amd as cdpr  # noqa: E402


def main():
    robots = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    cfg = cdpr.Config(batch=robots)  # the shipped 4-cable robot and launch-file gains
    plugin = cdpr.CdprGazeboPlugin()
    plugin.Load(cfg)
    poses = []
    plugin.bus.subscribe("platformPose", poses.append)
    stimulus = cdpr.stimulus.sine_velocity(cfg.n_cables)  # amp 0.05 m/s, 0.1 Hz, 100 Hz publisher
    for k in range(int(seconds * 100)):
        plugin.bus.publish("jointVelocities", cdpr.Joy(axes=next(stimulus)))
        plugin.update(10)  # ten 1 ms world steps per command sample
        if k % 50 == 49:
            p = poses[-1]
            print(f"t = {p.header.stamp:6.3f} s  platform z = {p.pose.position[0, 2]:.6f} m  vz = {p.velocity.linear[0, 2]:+.6f} m/s")
    q, qd, eff = plugin.engine.joint_states()
    print("cable positions [m]:", np.round(q[0], 6), " efforts [N]:", np.round(eff[0], 4))


if __name__ == "__main__":
    main()
