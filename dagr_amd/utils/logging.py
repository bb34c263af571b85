"""The two helpers of ``src/dagr/utils/logging.py`` the test scripts call (:101-117), without wandb (not part of this
stack): the run directory ``<output>/<dataset>/<task>/<exp_name>`` and a printed hyper-parameter table."""
from pathlib import Path, PurePath
from pprint import pprint


def set_up_logging_directory(dataset, task, output_directory, exp_name="temp"):
    out = Path(output_directory) / dataset / task / exp_name
    out.mkdir(parents=True, exist_ok=True)
    return out


def log_hparams(args):
    pprint({k: str(v) if isinstance(v, PurePath) else v for k, v in vars(args).items()})


def log_bboxes(*args, **kwargs):
    """Image logging of boxes (logging.py:119-213) is visualisation, outside the hot path: no-op."""
    return None
