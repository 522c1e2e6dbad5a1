"""Convert a spconv-1 (HAIS) checkpoint to the spconv-2 layout and SoftGroup key names.
Same command line and output file as the reference's tools/convert_checkpoint.py:
    python tools/convert_checkpoint.py CHECKPOINT.pth   ->   CHECKPOINT_spconv2.pth"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd.util.checkpoint import convert_checkpoint_file  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('checkpoint', type=str, help='spconv1 checkpoint')
    print(convert_checkpoint_file(ap.parse_args().checkpoint))
