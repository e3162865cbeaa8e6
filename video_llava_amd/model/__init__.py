from .video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VideoChatGPTLlamaModel, VisionConfig  # noqa: F401
