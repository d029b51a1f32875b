"""opticalflow_amd - MI355X-native variational optical flow (drop-in for the hot path of
kursawe/OpticalFlow's ``source/optical_flow.py::variational_optical_flow``)."""
from .optical_flow import variational_optical_flow, vary_regularisation, make_fake_data_frame, blur_movie  # noqa: F401

__version__ = "0.1.0"
