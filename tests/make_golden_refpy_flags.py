"""Golden namespaces of the reference's own ``FLAGS()`` (``/root/reference/src/dagr/utils/args.py:54-110``, imported and run
here; it needs only argparse + yaml) on the five command lines its readme documents (readme.md:68-75, 107-113, 131-138,
168-171, 180-184), with ``$LOG_DIR`` / ``$DSEC_ROOT`` / ``$DAGR_DIR`` spelled as literal placeholder paths.  Two of the
readme's lines name ``config/eagr-s-dsec.yaml``, a file the reference's tree does not hold: the golden for those is made
with ``dagr-s-dsec.yaml`` (what this repo's scripts resolve the name to, with a notice).

  python tests/make_golden_refpy_flags.py      ->  tests/golden/ref_py_flags.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# script -> argv as the readme gives it ("config/..." is relative to the reference's root there, to this repo's root here)
README_LINES = {
    "run_test_interframe.py@readme:68-75": ["--config", "config/dagr-s-dsec.yaml", "--use_image", "--img_net", "resnet50",
                                            "--checkpoint", "data/dagr_s_50.pth", "--batch_size", "8",
                                            "--dataset_directory", "data/DSEC_fragment", "--no_eval",
                                            "--output_directory", "/LOG_DIR"],
    "run_test.py@readme:107-113": ["--config", "config/dagr-s-dsec.yaml", "--use_image", "--img_net", "resnet50",
                                   "--checkpoint", "data/dagr_s_50.pth", "--batch_size", "8",
                                   "--dataset_directory", "/DSEC_ROOT", "--output_directory", "/LOG_DIR"],
    "run_test_interframe.py@readme:131-138": ["--config", "config/eagr-s-dsec.yaml", "--use_image", "--img_net", "resnet50",
                                              "--checkpoint", "data/dagr_s_50.pth", "--batch_size", "8",
                                              "--dataset_directory", "/DSEC_ROOT", "--output_directory", "/LOG_DIR",
                                              "--num_interframe_steps", "10"],
    "train_ncaltech101.py@readme:168-171": ["--config", "config/dagr-l-ncaltech.yaml", "--exp_name", "ncaltech_l",
                                            "--dataset_directory", "/DAGR_DIR/data/", "--output_directory",
                                            "/DAGR_DIR/logs/"],
    "train_dsec.py@readme:180-184": ["--config", "config/dagr-s-dsec.yaml", "--exp_name", "dsec_s_50",
                                     "--dataset_directory", "/DAGR_DIR/data/", "--output_directory", "/DAGR_DIR/logs/",
                                     "--use_image", "--img_net", "resnet50", "--batch_size", "32"],
}


def main():
    sys.path.insert(0, "/root/reference/src")             # (only here: importing this file for README_LINES must not shadow `dagr`)
    import dagr.utils.args as rargs                       # the reference's module
    out = {}
    argv0, cwd0 = list(sys.argv), os.getcwd()
    os.chdir("/root/reference")                           # the readme runs its commands from the reference's root
    for key, argv in README_LINES.items():
        argv = [a.replace("eagr-", "dagr-") for a in argv]
        sys.argv = ["x"] + argv
        ns = rargs.FLAGS()
        out[key] = {k: (str(v) if not isinstance(v, (int, float, bool, str)) else v) for k, v in vars(ns).items()}
    sys.argv = argv0
    os.chdir(cwd0)
    path = os.path.join(ROOT, "tests", "golden", "ref_py_flags.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
