"""`python -m neuma_amd.render -c <config.yaml> -vn <video name> ...`: the reference's experiments/render.py entry point."""
import sys

from ..evaluate import main

if __name__ == "__main__":
    sys.exit(main())
