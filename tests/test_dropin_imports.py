"""Drop-in import surface: with `video-gpt_amd/dropin` on sys.path every `from LVM... import ...` line of the reference's
entry scripts that names an in-scope symbol resolves to the MI355X implementation, with the reference's signatures.

Lines performed (reference file:line):
  LVM/inference/LVM_video_frameblock_autoregressive_inference.py:19  from LVM import LVMPipeline
  ...:20                                                             from LVM.acceleration.parallel_states import init_npu_env, hccl_info
  ...:26 (NV branch)                                                 from LVM.transform.sdpa_transform import replace_attention
  LVM/train/train_x1_stage1_noiseinput.py:40                         from LVM.train_helper.loss import is_all_equal
  ...:41                                                             from LVM import LVMTraining, LVMProcessor
  ...:45                                                             from LVM.transform.sdpa_transform import replace_attention as replace_simple_attention
  ...:46                                                             from LVM.acceleration.parallel_states import initialize_sequence_parallel_state, hccl_info
  ...:47                                                             from LVM.train_helper import TrainDataCollator          (DatasetFromVideo: video I/O, out of scope)
  ...:48                                                             from LVM.train_helper import training_losses_x1_noise_input
  ...:49-56                                                          from LVM.utils import vae_encode                       (logging / EMA / crop helpers: out of scope)
  LVM/train/train_x1_stage2_noiseinput_cp.py                          from LVM import LVMTraining_CP; TrainDataCollator_FrameBlock
Out of scope by SURVEY.md §2: decord / accelerate / deepspeed / peft imports of those scripts."""
import importlib
import inspect
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dropin():
    importlib.import_module("video-gpt_amd")
    path = os.path.join(ROOT, "video-gpt_amd", "dropin")
    saved = {k: v for k, v in sys.modules.items() if k == "LVM" or k.startswith("LVM.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, path)
    yield
    sys.path.remove(path)
    for k in [k for k in sys.modules if k == "LVM" or k.startswith("LVM.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_import_lines_resolve(dropin):
    from LVM import LVMPipeline
    from LVM.acceleration.parallel_states import init_npu_env, hccl_info
    from LVM.transform.sdpa_transform import replace_attention
    from LVM.train_helper.loss import is_all_equal
    from LVM import LVMTraining, LVMProcessor
    from LVM.transform.sdpa_transform import replace_attention as replace_simple_attention
    from LVM.acceleration.parallel_states import initialize_sequence_parallel_state, hccl_info as h2
    from LVM.train_helper import TrainDataCollator, TrainDataCollator_FrameBlock
    from LVM.train_helper import training_losses_x1_noise_input
    from LVM.utils import vae_encode
    from LVM import LVMTraining_CP, LVMScheduler, LVM as LVMModel
    from LVM.pipeline import LVMPipeline as P2
    from LVM.model import LVM as M2, LVMTraining as T2
    from LVM.processor import LVMProcessor as Pr2, LVMCollator
    from LVM.scheduler import LVMScheduler as S2
    pkg = importlib.import_module("video-gpt_amd")
    assert LVMPipeline is P2 is importlib.import_module("video-gpt_amd.pipeline").LVMPipeline
    assert LVMTraining is T2 and LVMModel is M2 and LVMProcessor is Pr2 and LVMScheduler is S2
    assert hccl_info is h2 and replace_attention is replace_simple_attention
    assert is_all_equal([torch.ones(2), torch.ones(2)]) and not is_all_equal([torch.ones(2), torch.zeros(2)])
    # signatures: argument names / order of the reference
    names = lambda f: list(inspect.signature(f).parameters)
    assert names(LVMPipeline.__call__)[:19] == [
        "self", "input_images", "height", "width", "gen_num", "num_inference_steps", "use_img_guidance", "img_guidance_scale",
        "max_input_image_size", "offload_model", "use_kv_cache", "offload_kv_cache", "use_input_image_size_as_output", "dtype",
        "seed", "output_type", "time_shifting_factor", "prediction_type", "clean_image_noise_level"]      # LVM/pipeline.py:138-158
    assert names(LVMPipeline.prompt_condition_frame_block_autoregressive_inference)[:20] == [
        "self", "input_images", "height", "width", "gen_nums", "num_inference_steps", "use_img_guidance", "img_guidance_scale",
        "max_input_image_size", "offload_model", "use_kv_cache", "offload_kv_cache", "use_input_image_size_as_output", "dtype",
        "seed", "output_type", "time_shifting_factor", "prediction_type", "clean_image_noise_level", "max_frame_window"]   # :347-368
    assert names(training_losses_x1_noise_input)[:11] == ["model", "x1", "model_kwargs", "snr_type", "patch_weight",
                                                          "input_noise", "cls_weight", "order", "frame_blocks", "exp_time",
                                                          "device"]                                   # loss.py:128-140
    assert names(LVMTraining.forward)[:10] == ["self", "x", "timestep", "input_ids", "input_img_latents", "input_image_sizes",
                                               "attention_mask", "position_ids", "denoise_image_sizes", "time_emb_inx"]
    assert names(TrainDataCollator.__init__)[1:8] == ["pad_token_id", "hidden_size", "keep_raw_resolution", "frame_num",
                                                      "sequence_parallel_size", "batch_size", "block_aware"]   # data.py:405-412
    assert names(vae_encode)[:5] == ["vae", "x", "weight_dtype", "seed", "batch_encode"]               # LVM/utils.py:99
    assert names(init_npu_env) == ["args"] and names(initialize_sequence_parallel_state) == ["sequence_parallel_size"]


def test_parallel_states_single_process(dropin, monkeypatch):
    """init_npu_env at WORLD_SIZE 1 (how the inference script runs on one GPU, inference.py:44): no process group is
    needed, hccl_info describes a sequence-parallel group of one."""
    from LVM.acceleration.parallel_states import init_npu_env, hccl_info, get_sequence_parallel_state
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "1")
    args = init_npu_env(types.SimpleNamespace(sequence_parallel_size=1))
    assert (args.local_rank, args.world_size) == (0, 1)
    assert (hccl_info.world_size, hccl_info.rank, hccl_info.group) == (1, 0, None) and get_sequence_parallel_state()


def test_train_collators_match_reference_index_conventions(dropin):
    """TrainDataCollator / TrainDataCollator_FrameBlock (LVM/train_helper/data.py:404-537) on features laid out by the
    processor; live against the reference's own classes when its checkout is present."""
    from LVM import LVMProcessor
    from LVM.train_helper import TrainDataCollator, TrainDataCollator_FrameBlock
    P = importlib.import_module("video-gpt_amd.processor")
    proc = LVMProcessor(P.SpecialTokenizer(10, 11, 12))
    F = 3
    prompt = "".join(f"<|diffusion|><|image_{i + 1}|><img><|image_{i + 1}|></img>" if i < F - 1 else f"<|diffusion|><|image_{i + 1}|>"
                     for i in range(F))
    feat = proc.process_multi_modal_prompt_training(prompt, [torch.full((3, 64, 64), float(i)) for i in range(F)])
    out = TrainDataCollator(2, 8, True, F, batch_size=2)([feat])
    assert out["input_ids"].shape[0] == 2                              # the lone sample is repeated to the batch size
    assert len(out["output_images"]) == 2 * F and len(out["input_pixel_values"]) == 2 * (F - 1)
    assert out["time_emb_inx"][0] == [s[0] - 1 for s in out["denoise_image_sizes"][0]]
    fbs = [1, 2, 1]
    prompt2, i, j, n = "", 0, 0, 0
    for k, fb in enumerate(fbs):
        for _ in range(fb):
            prompt2 += f"<|diffusion|><|image_{i + 1}|>"; i += 1; n += 1
        if k != len(fbs) - 1:
            for _ in range(fb):
                prompt2 += f"<img><|image_{j + 1}|></img>"; j += 1
    feat2 = proc.process_multi_modal_prompt_frame_block_training(prompt2, [torch.full((3, 64, 64), float(q)) for q in range(n)], fbs)
    feat2["frame_blocks"] = fbs
    n_slots = len(feat2["image_sizes"])
    out2 = TrainDataCollator_FrameBlock(2, 8, True, n_slots, batch_size=1)([feat2])
    assert len(out2["denoise_image_sizes"][0]) == sum(fbs) and len(out2["input_image_sizes"][0]) == sum(fbs[:-1])
    from oracle import extract_reference as X
    if X.available():
        import copy
        ns = X.collator_classes()
        data = X._extract("LVM/train_helper/data.py", ["TrainDataCollator", "TrainDataCollator_FrameBlock"],
                          extra_ns={"LVMCollator": ns.LVMCollator, "copy": copy})
        ref = data.TrainDataCollator(2, 8, True, F, batch_size=2)([copy.deepcopy(feat)])
        ref2 = data.TrainDataCollator_FrameBlock(2, 8, True, n_slots, batch_size=1)([copy.deepcopy(feat2)])
        for a, b in ((out, ref), (out2, ref2)):
            for k in ("input_ids", "position_ids"):
                assert torch.equal(a[k], b[k])
            assert torch.equal(a["attention_mask"].bool(), b["attention_mask"].bool())
            for k in ("input_image_sizes", "denoise_image_sizes", "time_emb_inx"):
                assert {kk: [list(x) if isinstance(x, (list, tuple)) else x for x in v] for kk, v in a[k].items()} == \
                       {kk: [list(x) if isinstance(x, (list, tuple)) else x for x in v] for kk, v in b[k].items()}
            for k in ("input_pixel_values", "output_images"):
                assert len(a[k]) == len(b[k]) and all(torch.equal(u, v) for u, v in zip(a[k], b[k]))
