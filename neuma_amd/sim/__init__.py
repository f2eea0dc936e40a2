from .abstract import State, Model, ModelBuilder, StateInitializer, StaticsInitializer
from .mpm import (MPMStatics, MPMParticleData, MPMConstant, MPMState, MPMModel, MPMModelBuilder, MPMInitData,
                  MPMStateInitializer, MPMStaticsInitializer)
from .interface import MPMSimFunction, MPMSim, MPMDiffSim, MPMCacheDiffSim, MPMForwardSim, MPMExtraSim
