"""Token strings of the PG-Video-LLaVA prompt protocol (reference: video_chatgpt/constants.py:8-12,
video_chatgpt/inference.py:6-10).  These literals are part of the checkpoint/tokenizer contract."""
DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_VIDEO_PATCH_TOKEN = "<vid_patch>"
DEFAULT_VID_START_TOKEN = "<vid_start>"
DEFAULT_VID_END_TOKEN = "<vid_end>"
DEFAULT_TRANSCRIPT_START = "The noisy audio transcript of this video is:"

NUM_TEMPORAL_TOKENS = 100   # hard-coded in the reference: inference.py:31, chat.py:80, eval/model_utils.py:148
CLIP_WIDTH = 1024           # hard-coded at model/video_chatgpt.py:106 and save_spatio_temporal_clip_features.py:109
SELECT_HIDDEN_STATE_LAYER = -2   # inference.py:94
