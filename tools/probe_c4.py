import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.probe_neutra import run
if __name__ == '__main__':
    for T in [int(v) for v in os.environ.get('C4_T', '2').split(',')]:
        run(65536, 128, 128, 2, 10, T)
