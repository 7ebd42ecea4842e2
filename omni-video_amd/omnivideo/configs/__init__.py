"""Model / sampling configuration values of the reference (omnivideo/configs/*.py), kept as
attribute-style dicts so `config.num_train_timesteps` and `config.get('boundary', ...)`
both work as they do with the reference's EasyDict objects."""
import torch


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _shared():
    # reference configs/shared_config.py:8-20
    return Config(
        t5_model="umt5_xxl", t5_dtype=torch.bfloat16, text_len=512, param_dtype=torch.bfloat16,
        num_train_timesteps=1000, sample_fps=16, frame_num=81,
        sample_neg_prompt_en=(
            "overly vivid colors, overexposed, static, blurry and unclear details, subtitles, text overlays, "
            "stylized artwork, painting, illustration, still image, frozen frame, grayish overall tone, worst "
            "quality, low quality, JPEG compression artifacts, ugly, incomplete, extra fingers, poorly drawn "
            "hands, poorly drawn face, deformed, disfigured, malformed anatomy, distorted limbs, fused fingers, "
            "motionless frame, cluttered background, three legs, walking backward, expression drift, teeth "
            "flickering, uneven facial muscles, facial shivering"))


def _common(cfg):
    cfg.update(t5_checkpoint="models_t5_umt5-xxl-enc-bf16.pth", t5_tokenizer="google/umt5-xxl",
               vae_checkpoint="Wan2.1_VAE.pth", vae_stride=(4, 8, 8), patch_size=(1, 2, 2), freq_dim=256,
               window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6, sample_shift=12.0,
               sample_steps=40, boundary=0.875, sample_guide_scale=(3.0, 4.0), use_visual_context_adapter=True,
               visual_context_adapter_patch_size=(1, 4, 4), condition_mode="full", vlm_in_dim=2048)
    return cfg


# reference configs/wan_t2v_1_3B.py:19-40
t2v_1_3B = _common(_shared())
t2v_1_3B.update(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)
# reference configs/wan_t2v_A14B.py:19-42
t2v_A14B = _common(_shared())
t2v_A14B.update(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, low_noise_checkpoint="low_noise_model",
                high_noise_checkpoint="high_noise_model")

WAN_CONFIGS = {"t2v-A14B": t2v_A14B, "t2v-1.3B": t2v_1_3B}
SIZE_CONFIGS = {"720*1280": (720, 1280), "1280*720": (1280, 720), "480*832": (480, 832), "832*480": (832, 480)}
MAX_AREA_CONFIGS = {k: v[0] * v[1] for k, v in SIZE_CONFIGS.items()}
SUPPORTED_SIZES = {"t2v-A14B": tuple(SIZE_CONFIGS), "t2v-1.3B": tuple(SIZE_CONFIGS)}
