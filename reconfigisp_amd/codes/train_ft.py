#!/usr/bin/env python3
"""Search with online proxy fine-tuning (reference train_ft.py): same command line as train.py, model
``darts_ft``; ``proxy_ft_params.ft_interval`` sets how often ``finetune_proxies()`` runs."""
import os
import sys

if __package__ in (None, ''):
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir, os.pardir)))
    __package__ = 'reconfigisp_amd.codes'

from .train import main

if __name__ == '__main__':
    main()
