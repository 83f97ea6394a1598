"""predict_pv_yield_amd — MI355X-native hot path of openclimatefix/predict_pv_yield.

Optical-flow advection of satellite tiles (Farnebäck + cv.remap semantics) and the Conv3D PV-yield
model (forward / backward / Adam) as hand-written gfx950 kernels behind a C ABI
(include/pv_yield_hip.h), driven through the reference's LightningModule / DataModule / Hydra surface.
"""
__version__ = "0.1.0"
