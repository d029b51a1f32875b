#!/usr/bin/env python3
"""The reference's only enabled experiment (analysis/analyse_variational_optical_flow.py:26-66,
``simple_test_with_data_on_boundary``) run against the MI355X-native drop-in module: two 50x50 frames of a
translating Gaussian hat (true v = (0.1, 0.2), remodelling rate 0.05), same calls, same printed summary.  The
six-panel overlay movie of the original is written as well (``--movie``; .gif through pillow, the reference writes
.mp4 where ffmpeg is installed).

    python examples/simple_test_with_data_on_boundary.py [--movie]        (needs an MI355X)
"""
import os
import sys

import numpy as np

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "source"))
import optical_flow  # noqa: E402  (the drop-in module, same import the reference's scripts use)


def simple_test_with_data_on_boundary():
    v_x = 0.1
    v_y = 0.2
    first_frame, delta_x = optical_flow.make_fake_data_frame(x_position=2.5, y_position=2.5, sigma=3, width=5,
                                                             dimension=50, include_noise=False)
    second_frame, _ = optical_flow.make_fake_data_frame(x_position=2.5 + v_x, y_position=2.5 + v_y, sigma=3, width=5,
                                                        dimension=50, include_noise=False)
    second_frame += 0.05
    movie = np.stack((first_frame, second_frame))
    result = optical_flow.variational_optical_flow(movie, delta_x=delta_x, delta_t=1.0, speed_alpha=1.0,
                                                   remodelling_alpha=10000.0, smoothing_sigma=None)
    if "--movie" in sys.argv:
        import matplotlib
        matplotlib.use("Agg")
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "output")
        os.makedirs(out, exist_ok=True)
        optical_flow.make_joint_overlay_movie(result, os.path.join(out, "simple_example_joint_result.gif"), autoscale=True,
                                              arrow_scale=0.5, arrow_boxsize=4, dpi=100)
    print('mean and max final v_x are')
    print(np.mean(result['v_x']))
    print(np.max(result['v_x']))
    print('mean and max final v_y are')
    print(np.mean(result['v_y']))
    print(np.max(result['v_y']))
    print('mean and max final remodelling are')
    print(np.mean(result['remodelling']))
    print(np.max(result['remodelling']))
    print('converged:', result['converged'])
    return result


if __name__ == "__main__":
    simple_test_with_data_on_boundary()
