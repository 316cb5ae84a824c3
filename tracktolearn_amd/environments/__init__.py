from tracktolearn_amd.environments.env import BaseEnv  # noqa: F401
from tracktolearn_amd.environments.tracking_env import TrackingEnvironment  # noqa: F401
from tracktolearn_amd.environments.noisy_tracking_env import NoisyTrackingEnvironment  # noqa: F401
from tracktolearn_amd.environments.stopping_criteria import StoppingFlags  # noqa: F401
