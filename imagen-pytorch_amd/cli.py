"""`imagen sample` — the reference's sampling command line (imagen_pytorch/cli.py:27-64) over the MI355X sampler:

    python -m imagen_pytorch_amd.cli sample --model ./imagen.pt --cond_scale 5 "a prompt"

loads a checkpoint saved by the reference's trainer (config + weights, EMA unets when present), samples one image for the prompt and
writes `./<slugified prompt>.png`.  The `config` and `train` commands of the reference are training-side and not part of this build.
The prompt goes through the model's `encode_text` hook (T5 from local files, t5.py) — it fails with an actionable message when the
encoder weights are not on disk.
"""
from __future__ import annotations

from pathlib import Path

import click
import torch


def simple_slugify(text: str, max_length: int = 255) -> str:
    """cli.py:18-19."""
    return text.replace('-', '_').replace(',', '').replace(' ', '_').replace('|', '--').strip('-_./\\')[:max_length]


@click.group()
def imagen():
    pass


@imagen.command(help='Sample from the Imagen model checkpoint')
@click.option('--model', default='./imagen.pt', help='path to trained Imagen model')
@click.option('--cond_scale', default=5, help='conditioning scale (classifier free guidance) in decoder')
@click.option('--load_ema', default=True, help='load EMA version of unets if available')
@click.option('--trust_checkpoint', is_flag=True, default=False,
              help='allow the full unpickler for checkpoints that hold more than tensors (executes code stored in the file)')
@click.argument('text')
def sample(model, cond_scale, load_ema, trust_checkpoint, text):
    from .checkpoint import _load_checkpoint_file, load_imagen_from_checkpoint

    model_path = Path(model)
    full_model_path = str(model_path.resolve())
    assert model_path.exists(), f'model not found at {full_model_path}'
    version = _load_checkpoint_file(model_path, trust_checkpoint).get('version')
    print(f'loading Imagen from {full_model_path}, saved at version {version}')
    model_obj = load_imagen_from_checkpoint(str(model_path), load_ema_if_available=load_ema, trust_checkpoint=trust_checkpoint)
    model_obj.to(torch.device('cuda'))
    pil_image = model_obj.sample([text], cond_scale=cond_scale, return_pil_images=True)
    image_path = f'./{simple_slugify(text)}.png'
    pil_image[0].save(image_path)
    print(f'image saved to {str(image_path)}')


def main():
    imagen()


if __name__ == '__main__':
    main()
