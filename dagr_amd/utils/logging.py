"""The helpers of ``src/dagr/utils/logging.py`` the scripts call, without wandb (not part of this stack): the run
directory ``<output>/<dataset>/<task>/<exp_name>`` (:101-110), a printed hyper-parameter table (:113-117) and the
``Checkpointer`` of the training scripts (:14-98; same file layout: ``last_model.pth`` / ``best_model_mAP_<x>.pth`` holding
``ema``, ``ema_updates``, ``model``, ``optimizer``, ``scheduler``, ``epoch``, ``args`` -- what ``run_test.py:57`` loads)."""
from pathlib import Path, PurePath
from pprint import pprint

import torch


def set_up_logging_directory(dataset, task, output_directory, exp_name="temp"):
    out = Path(output_directory) / dataset / task / exp_name
    out.mkdir(parents=True, exist_ok=True)
    return out


def log_hparams(args):
    pprint({k: str(v) if isinstance(v, PurePath) else v for k, v in vars(args).items()})


def log_bboxes(*args, **kwargs):
    """Image logging of boxes (logging.py:119-213) is visualisation, outside the hot path: no-op."""
    return None


class Checkpointer:
    def __init__(self, output_directory=None, args=None, optimizer=None, scheduler=None, ema=None, model=None):
        self.optimizer, self.scheduler, self.ema, self.model = optimizer, scheduler, ema, model
        self.output_directory = None if output_directory is None else Path(output_directory)
        self.args = args
        self.mAP_max = 0
        self.metric_log = []          # (epoch, metrics) in place of wandb.log

    @staticmethod
    def _mAP_of(path):
        return float(Path(path).name.split("_")[-1].split(".pth")[0])

    def search_for_checkpoint(self, folder, best=False):
        folder = Path(folder)
        found = sorted(folder.glob("*.pth"))
        last = folder / "last_model.pth"
        if not found:
            return None
        if not best and last in found:
            return last
        ranked = sorted((p for p in found if p != last), key=self._mAP_of)
        return ranked[-1] if ranked else None

    def restore_if_existing(self, folder, resume_from_best=False):
        if self.search_for_checkpoint(folder, best=resume_from_best) is None:
            return 0
        return self.restore_checkpoint(folder, best=resume_from_best)

    def restore_checkpoint(self, checkpoint_directory, best=False):
        path = self.search_for_checkpoint(checkpoint_directory, best)
        if path is None:
            raise FileNotFoundError(f"no checkpoint in {checkpoint_directory}")
        state = torch.load(path, map_location="cpu", weights_only=False)
        if self.ema is not None:
            self.ema.ema.load_state_dict(state.get("ema", state["model"]))
            self.ema.updates = state.get("ema_updates", 0)
        for target, key in ((self.model, "model"), (self.optimizer, "optimizer"), (self.scheduler, "scheduler")):
            if target is not None:
                target.load_state_dict(state[key])
        return state["epoch"]

    def checkpoint(self, epoch, name=""):
        self.output_directory.mkdir(exist_ok=True, parents=True)
        torch.save({"ema": self.ema.ema.state_dict(), "ema_updates": self.ema.updates, "model": self.model.state_dict(),
                    "optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict(), "epoch": epoch,
                    "args": self.args}, self.output_directory / f"{name}.pth")

    def process(self, data, epoch):
        """Keep the best-mAP checkpoint (logging.py:90-98)."""
        self.metric_log.append((epoch, dict(data)))
        mAP = data["mAP"]
        if mAP > self.mAP_max:
            self.checkpoint(epoch, name=f"best_model_mAP_{mAP}")
            self.mAP_max = mAP
