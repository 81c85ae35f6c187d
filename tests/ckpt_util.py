"""Test helper: write a tiny DiffewS checkpoint directory in the diffusers layout the launcher loads
(evaluation_util/main_oss.py:338-369): unet/ vae/ scheduler/ text_encoder/ tokenizer/ -- synthetic
weights, a 2-layer CLIP text tower whose hidden size equals the UNet's cross_attention_dim."""
import json
import os


def make_checkpoint_dir(root, dtype=None, seed=0):
    import torch
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    from diffews_amd import config, weights
    root = str(root)
    ucfg, vcfg = config.get("tiny_unet"), config.get("tiny_vae")
    usd = weights.synthetic_unet_state_dict(ucfg, round_to=dtype)
    vsd = weights.synthetic_vae_state_dict(vcfg, round_to=dtype)
    weights.save_pretrained(root, ucfg, usd, "unet")
    weights.save_pretrained(root, vcfg, vsd, "vae")
    os.makedirs(os.path.join(root, "scheduler"), exist_ok=True)
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(config.get("scheduler"), f)
    torch.manual_seed(seed)
    tcfg = CLIPTextConfig(vocab_size=64, hidden_size=ucfg["cross_attention_dim"], intermediate_size=64,
                          num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
                          bos_token_id=62, eos_token_id=63, pad_token_id=0)
    enc = CLIPTextModel(tcfg).eval()
    enc.save_pretrained(os.path.join(root, "text_encoder"))
    td = os.path.join(root, "tokenizer")
    os.makedirs(td, exist_ok=True)
    vocab = {chr(97 + i) + "</w>": i for i in range(26)}
    vocab.update({chr(97 + i): 26 + i for i in range(26)})
    vocab.update({f"<|pad{i}|>": i for i in range(52, 62)})
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = 62, 63
    with open(os.path.join(td, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(td, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    CLIPTokenizer(os.path.join(td, "vocab.json"), os.path.join(td, "merges.txt"), model_max_length=77).save_pretrained(td)
    with torch.no_grad():
        embed = enc(torch.tensor([[62, 63]]))[0].float()      # CLIP("") = [BOS, EOS] (P:591-600)
    return dict(root=root, ucfg=ucfg, vcfg=vcfg, usd=usd, vsd=vsd, text_embed=embed)
