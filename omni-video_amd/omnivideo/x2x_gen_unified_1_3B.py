"""Single-model 1.3B variant of the pipeline (reference: omnivideo/x2x_gen_unified_1_3B.py):
one diffusion model, `high_noise_model` / `low_noise_model` kept as aliases (:139-140), no
expert switch at the noise boundary (:186-195)."""
from .x2x_gen_unified import OmniVideoX2XUnified


class OmniVideoX2XUnified1_3B(OmniVideoX2XUnified):

    def _build_models(self, wan_config, checkpoint_dir, vlm_in_dim, use_vca, vca_patch, max_context_len):
        # reference :107-136; without wan_config the DiT comes from checkpoint_dir/<config.checkpoint>
        self.model = self._new_model(wan_config, vlm_in_dim, use_vca, vca_patch, max_context_len, checkpoint_dir,
                                     self.config.get("checkpoint", None))
        self.high_noise_model = self.model
        self.low_noise_model = self.model

    def _prepare_model_for_timestep(self, t, boundary, offload_model=False):
        return self.model
