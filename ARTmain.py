"""`python ARTmain.py CONFIG_FILE` and `from ARTmain import main` as in the reference; the implementation lives
in attosecondraytracing_amd/ARTmain.py."""
import sys

from attosecondraytracing_amd.ARTmain import (complete_defaults, load_config, main, make_plots,  # noqa: F401
                                              optimize_detector, print_banner, run_ART, setup_detector, cli)

if __name__ == "__main__":
    sys.exit(cli())
