"""Import shim: the reference's scripts do ``sys.path.append(<repo>/source); import optical_flow``
(analysis/analyse_variational_optical_flow.py:22-23).  This module exposes the MI355X-native hot
path under the same module name."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opticalflow_amd.optical_flow import *  # noqa: F401,F403,E402
from opticalflow_amd.optical_flow import (variational_optical_flow, vary_regularisation, make_fake_data_frame, blur_movie,  # noqa: F401,E402
                                          format_elapsed_time, apply_constant_boundary_condition,
                                          subsample_velocities_for_visualisation, costum_imshow,
                                          make_velocity_overlay_movie, make_joint_overlay_movie)
