#!/usr/bin/env python
"""Twin of the reference's ``scripts/train_dsec.py`` (:41-184): the training loop of ``train_ncaltech101.py`` (same file
here: optimizer, schedule, EMA, checkpointer, data-parallel replicas over RCCL) on the DSEC settings -- two head scales,
flip / zoom / translate augmentations, ``DSEC(root / "dsec", "train" | "val", ..., min_bbox_diag=15, min_bbox_height=10)``
(:125-128) and, with ``--use_image --img_net resnet50``, the image branch: its features enter the graph detached, its
CNN head trains on its own against the earlier frame's boxes (dagr.py:241-268).  Without ``--dataset_directory`` the
run is on labelled synthetic samples at DSEC's half resolution (320 x 215, with frames when ``--use_image``).

  python scripts/train_dsec.py --config dagr-s --use_image --img_net resnet18 --epochs 2 --samples 64 --batch_size 8
"""
import train_ncaltech101 as T


def main(argv=None, model_factory=None):
    return T.main(argv, model_factory=model_factory, preset="dsec")


if __name__ == "__main__":
    main()
