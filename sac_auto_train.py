#!/usr/bin/env python3
"""Entry point: tracktolearn_amd.trainers.sac_auto_train.main (the reference
runs TrackToLearn/trainers/sac_auto_train.py)."""
from tracktolearn_amd.trainers.sac_auto_train import main

if __name__ == '__main__':
    main()
