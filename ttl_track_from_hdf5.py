#!/usr/bin/env python3
"""Console entry point (setup.py:88-91 of the reference installs
``ttl_track_from_hdf5.py``): tracktolearn_amd.runners.ttl_track_from_hdf5.main."""
from tracktolearn_amd.runners.ttl_track_from_hdf5 import main

if __name__ == '__main__':
    main()
