#!/usr/bin/env python3
"""Drop-in for the reference's scripts/save_spatio_temporal_clip_features.py (same CLI, same .pkl files); the work is done by
video_llava_amd.feature_extraction on the MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_llava_amd.feature_extraction import get_seq_frames, get_spatio_temporal_features, load_video, main, parse_args  # noqa: E402,F401

if __name__ == "__main__":
    main()
