"""bayhunter_amd -- MI355X-native forward-model + likelihood engine behind BayHunter's plugin
surface.  Importing this package does not touch the GPU; the native library is
loaded when the first `Engine` is created."""
from .engine import Engine, EngineError, default_engine, pack_models  # noqa: F401
from .Models import Model, ModelMatrix  # noqa: F401
from .surf96_modsw import SurfDisp  # noqa: F401
from .rfmini_modrf import RFminiModRF  # noqa: F401
from .Targets import (ObservedData, ModeledData, Valuation, SingleTarget, JointTarget,  # noqa: F401
                      RayleighDispersionPhase, RayleighDispersionGroup, LoveDispersionPhase,
                      LoveDispersionGroup, PReceiverFunction, SReceiverFunction, select_noise_laws)
from .chains import ChainBatch, MCMC_Optimizer  # noqa: F401
from .results import save_config, save_final_distribution, get_outliers  # noqa: F401


def __getattr__(name):  # DeviceChains needs torch: imported on first use only
    if name == "DeviceChains":
        from .device_chains import DeviceChains
        return DeviceChains
    raise AttributeError(name)


__version__ = "0.1.0"
