"""Prompt templates (host string work only).  Mirrors the public surface of the reference's
video_chatgpt/video_conversation.py -- `conv_templates[mode].copy()`, `.append_message`, `.get_prompt`,
`.roles`, `.sep`, `.sep2`, `.sep_style` -- as used by video_chatgpt/inference.py:77-80,101 and chat.py.
The system prompts and separators are protocol constants of the released checkpoints
(video_conversation.py:120-171); the prompt assembly rules follow :28-61 and are pinned against the
reference's own output in tests/golden/prompts.json.
"""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional, Sequence


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[Optional[str]]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    @staticmethod
    def _text(message):
        # a message may be a (text, video_path) tuple (video_conversation.py:34-35)
        return message[0] if isinstance(message, tuple) else message

    def get_prompt(self) -> str:
        if self.sep_style == SeparatorStyle.SINGLE:
            parts = [self.system, self.sep]
            for role, msg in self.messages:
                parts.append(f"{role}: {self._text(msg)}{self.sep}" if msg else f"{role}:")
            return "".join(parts)
        if self.sep_style == SeparatorStyle.TWO:
            seps = (self.sep, self.sep2)
            parts = [self.system, seps[0]]
            for i, (role, msg) in enumerate(self.messages):
                parts.append(f"{role}: {self._text(msg)}{seps[i % 2]}" if msg else f"{role}:")
            return "".join(parts)
        if self.sep_style == SeparatorStyle.MPT:
            parts = [self.system, self.sep]
            for role, msg in self.messages:
                parts.append(f"{role}{self._text(msg)}{self.sep}" if msg else role)
            return "".join(parts)
        raise ValueError(f"Invalid style: {self.sep_style}")

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_video_frames(self, n_clips=1, num_frm=100):
        from .eval.model_utils import load_video
        frames = []
        for i, (_role, msg) in enumerate(self.messages[self.offset:]):
            if i % 2 == 0 and isinstance(msg, tuple):
                frames.extend(load_video(msg[1], n_clips, num_frm))
        return frames

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[r, m] for r, m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2)

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": self.messages, "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2}


_VIDEO_ASSISTANT_TAIL = ("You are able to understand the video content that the user provides, and assist the user with a "
                         "variety of tasks using natural language."
                         "Follow the instructions carefully and explain your answers in detail based on the provided video.")

conv_v1_2 = Conversation(
    system="A chat between a curious human and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the human's questions.",
    roles=("Human", "Assistant"),
    messages=[["Human", "What are the key differences between renewable and non-renewable energy sources?"],
              ["Assistant", "Renewable energy sources are those that can be replenished naturally.\n"]],
    offset=2, sep_style=SeparatorStyle.SINGLE, sep="###")

conv_vicuna_v1_1 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

conv_video_chatgpt_v1 = Conversation(
    system="You are Video-ChatGPT, a large vision-language assistant. " + _VIDEO_ASSISTANT_TAIL,
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

conv_pg_video_llava = Conversation(
    system="You are PG-Video-LLaVA, a large vision-language assistant. " + _VIDEO_ASSISTANT_TAIL,
    roles=("USER", "ASSISTANT"), version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")

default_conversation = conv_v1_2
conv_templates = {
    "default": conv_v1_2,
    "video-chatgpt_v1": conv_video_chatgpt_v1,
    "vicuna_v1_1": conv_vicuna_v1_1,
    "pg-video-llava": conv_pg_video_llava,
}
