"""``utils`` under the reference's package name, so that the import block of the reference's entry scripts
(train_accel_gpu.py:14-17, infer_accel_gpu.py:14-17) resolves against this repository unmodified:

    from utils.training import get_param_norm, get_grad_norm, count_parameters, move_to
    from utils.config import training_config, get_model_config
    from utils.dataset import setup_data
    from utils.metrics import Alignment, Uniformity

Each module re-exports the native package's implementation (mca-paper_amd/{config,data,metrics}.py); utils/training.py holds
the four small host helpers of the step loop."""
