#!/usr/bin/env python3
"""Console entry point (setup.py:88-91 of the reference installs
``ttl_track.py``): tracktolearn_amd.runners.ttl_track.main."""
from tracktolearn_amd.runners.ttl_track import main

if __name__ == '__main__':
    main()
