"""Interactive chat entry point with the reference's class, methods and CLI
(video_chatgpt/chat.py:15-222 `VideoChatGPTInterface`, :336-367 `parse_args` / `__main__`).

Same conversation state machine (add_text -> answer), same prompt assembly (`<video>` replaced by the placeholder run once, :129-130),
same stop-string handling and post-processing.  Device work runs in libpgv.  Two deliberate differences:
  * the reference re-runs the CLIP tower on the whole clip for EVERY turn (:137-144); here the pooled features of the uploaded clip
    are computed once in `upload_video` and reused by every `answer()` of the conversation;
  * the reference re-runs the decoder over the WHOLE conversation every turn (:129-154); here a later turn prefills only the tokens behind the
    prefix the KV cache already holds (`generate(kv_reuse_key=...)` -> pgv_llm_prefill_append) -- same tokens, `last_timings["reused_tokens"]`;
  * `PGVideoLLaVA` (grounding: GroundingDINO / SAM / DEVA / RAM / OpenAI entity matching, :225-333) and the WhisperX transcript model are
    side stacks outside the hot path: `--with_grounding` and `--use_asr` raise NotImplementedError.
"""
from __future__ import annotations

import argparse

import torch

from .constants import DEFAULT_TRANSCRIPT_START, DEFAULT_VID_END_TOKEN, DEFAULT_VID_START_TOKEN, DEFAULT_VIDEO_PATCH_TOKEN
from .model.utils import KeywordsStoppingCriteria
from .video_conversation import SeparatorStyle, conv_templates, default_conversation


class VideoChatGPTInterface:
    def __init__(self, args_model_name, args_projection_path, use_asr=False, conv_mode="pg-video-llava", temperature=0.2,
                 max_output_tokens=1024, components=None, do_sample=True, reuse_kv=True) -> None:
        if use_asr:
            raise NotImplementedError("--use_asr needs the WhisperX transcript stack, which is outside this package's hot path")
        self.use_asr = use_asr
        self.conv_mode = conv_mode
        if components is None:
            from .eval.model_utils import initialize_model
            components = initialize_model(args_model_name, args_projection_path)
        model, vision_tower, tokenizer, image_processor, video_token_len = components
        self.tokenizer, self.image_processor, self.vision_tower, self.model = tokenizer, image_processor, vision_tower, model
        self.temperature = temperature
        self.max_new_tokens = max_output_tokens
        self.do_sample = do_sample
        self.reuse_kv = reuse_kv
        self.frame_size = (image_processor.crop_size["height"], image_processor.crop_size["width"])
        if self.model.get_model().vision_config.use_vid_start_end:
            self.replace_token = DEFAULT_VID_START_TOKEN + DEFAULT_VIDEO_PATCH_TOKEN * video_token_len + DEFAULT_VID_END_TOKEN
        else:
            self.replace_token = DEFAULT_VIDEO_PATCH_TOKEN * video_token_len
        self.clear_history()

    def clear_history(self):
        self.state = default_conversation.copy()
        self.video_features = None          # pooled [100 + P, 1024] features of the uploaded clip (computed once per clip)
        self.kv_key = None                  # names the uploaded clip: later turns of the same clip continue the KV cache of the earlier ones
        self.last_timings = {}
        self.video_path = None
        self.video_frames = None
        self.transcript_text = None
        self.first_run = True

    def upload_video(self, video_path):
        """Decode + sample + resize the clip (reference :62-75) and run the vision stage ONCE for the whole conversation."""
        from .feature_extraction import load_video
        from .inference import video_features
        if isinstance(video_path, str):
            frames = load_video(video_path, shape=self.frame_size, device_resize=True)
        elif hasattr(video_path, "shape"):                 # uint8 [T, H, W, 3] frames handed over directly
            frames = video_path
        else:
            raise NotImplementedError
        self.video_path = video_path if isinstance(video_path, str) else None
        self.video_frames = frames
        self.video_features = video_features(frames, self.vision_tower, self.image_processor)
        self.kv_key = object() if self.reuse_kv else None
        self.transcript_text = None

    def add_text(self, text, video_path):
        """Queue one user turn (reference :90-104): empty input without a clip is skipped; a turn is cut at 1536 characters, the first turn of a
        clip at 1200 and carries the `<video>` (and, with ASR, `<audio_transcript>`) markers and the clip path."""
        self.state.skip_next = (not text) and video_path is None      # (overwritten below exactly like the reference: the turn is queued either way)
        limit = 1200 if self.first_run else 1536
        turn = text[:limit]
        if self.first_run:
            if "<video>" not in turn:
                turn += "\n<video>"
            if self.use_asr:
                turn += "\n<audio_transcript>"
            turn = (turn, video_path)
            self.state = default_conversation.copy()
        user, assistant = self.state.roles[0], self.state.roles[1]
        self.state.append_message(user, turn)
        self.state.append_message(assistant, None)
        self.state.skip_next = False

    def answer(self):
        if self.state.skip_next:
            return
        if self.video_features is None:
            raise RuntimeError("upload_video() first")
        if self.first_run:                                   # re-root the first turn on the chosen template (:113-120)
            first_user_turn = self.state.messages[-2][1]
            self.state = conv_templates[self.conv_mode].copy()
            self.state.append_message(self.state.roles[0], first_user_turn)
            self.state.append_message(self.state.roles[1], None)
            self.first_run = False
        prompt = self.state.get_prompt()
        prompt = prompt.replace("<video>", self.replace_token, 1)
        prompt = prompt.replace("<audio_transcript>", f'{DEFAULT_TRANSCRIPT_START}\n"{self.transcript_text}"', 1)
        inputs = self.tokenizer([prompt])
        input_ids = torch.as_tensor(inputs.input_ids)
        stop_str = self.state.sep if self.state.sep_style != SeparatorStyle.TWO else self.state.sep2
        stopping_criteria = KeywordsStoppingCriteria([stop_str], self.tokenizer, input_ids)
        self.state.messages[-1][-1] = ""
        with torch.inference_mode():
            output_ids = self.model.generate(input_ids, video_spatio_temporal_features=self.video_features.unsqueeze(0),
                                             do_sample=self.do_sample, temperature=float(self.temperature),
                                             max_new_tokens=min(int(self.max_new_tokens), 1536), stopping_criteria=[stopping_criteria],
                                             kv_reuse_key=self.kv_key, timings=self.last_timings)
        n_in = input_ids.shape[1]
        n_diff = int((input_ids.to(output_ids.device) != output_ids[:, :n_in]).sum())
        if n_diff > 0:
            print(f"[Warning] {n_diff} output_ids are not the same as the input_ids")
        outputs = self.tokenizer.batch_decode(output_ids[:, n_in:], skip_special_tokens=True)[0].strip()
        if outputs.endswith(stop_str):
            outputs = outputs[:-len(stop_str)]
        output = self._post_process_code(outputs.strip())
        self.state.messages[-1][-1] += output
        return output

    def interact(self):
        """Terminal loop (reference :173-199): ask for a clip, then turns until an empty line (new conversation) or Ctrl-C (quit)."""
        print("Welcome to PG-Video-LLaVA !")
        clip = None
        try:
            while True:
                if clip is None:
                    clip = input("Please enter the video file path:   ")
                    self.upload_video(clip)
                text = input("USER>>")
                if text:
                    self.add_text(text, clip)
                    print("ASSISTANT>>", self.answer())
                else:                                        # empty line: drop the conversation and ask for the next clip
                    print("----------\n\n")
                    self.clear_history()
                    clip = None
        except KeyboardInterrupt:
            print("----------\nQUITTING...")

    def print_state(self):
        lines = [f"SYSTEM: {self.state.system}"]
        lines += [f"{role}: {msg[0] if type(msg) is tuple else msg}" for role, msg in self.state.messages]
        print("\n".join(lines) + "\n")

    @staticmethod
    def _post_process_code(code):
        """Undo the `\\_` escapes inside fenced code blocks when the fences pair up (reference :212-222)."""
        fence = "\n```"
        parts = code.split(fence)
        if len(parts) > 1 and len(parts) % 2 == 1:
            parts[1::2] = [inner.replace("\\_", "_") for inner in parts[1::2]]
        return fence.join(parts)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Demo")
    parser.add_argument("--model-name", type=str, required=True)
    parser.add_argument("--projection_path", type=str, required=True)
    parser.add_argument("--use_asr", action="store_true", help="Whether to use audio transcripts or not")
    parser.add_argument("--conv_mode", type=str, required=False, default="pg-video-llava")
    parser.add_argument("--with_grounding", action="store_true", help="Run with grounding module")
    return parser.parse_args(argv)


if __name__ == "__main__":
    args = parse_args()
    if args.with_grounding:
        raise NotImplementedError("the grounding stack (GroundingDINO / SAM / DEVA / RAM) is outside this package's hot path")
    VideoChatGPTInterface(args_model_name=args.model_name, args_projection_path=args.projection_path, use_asr=args.use_asr,
                          conv_mode=args.conv_mode).interact()
