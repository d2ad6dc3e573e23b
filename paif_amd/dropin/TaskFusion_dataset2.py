from paif_amd.TaskFusion_dataset2 import *  # noqa: F401,F403
from paif_amd.TaskFusion_dataset2 import Fusion_dataset, prepare_data_path  # noqa: F401
