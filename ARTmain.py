"""`python ARTmain.py CONFIG_FILE` and `from ARTmain import main` as in the reference; the implementation lives
in attosecondraytracing_amd/ARTmain.py."""
import sys

from attosecondraytracing_amd.ARTmain import (analyse_chain_list, complete_defaults, load_config, main,  # noqa: F401
                                              make_plots, optimize_detector, print_banner, run_ART, setup_detector, cli)

if __name__ == "__main__":
    sys.exit(cli())
