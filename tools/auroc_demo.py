"""BASELINE config 4 surrogate: tf_efficientdet_d4 at 1024 px, soft-NMS path, image-level OOD score
max_a(-energy_a) for an "in-distribution" set vs an "OOD" set, AUROC computed on the device.

There is no COCO / OpenImages data (and no trained checkpoint) in this environment, so both sets are synthetic and the
weights are the seeded random initialisation of bench.py: the number demonstrates the pipeline (model -> per-anchor
energy -> image score -> AUROC, all in HIP kernels), not detection quality.  in-dist = smooth low-frequency images,
OOD = white noise.  Prints one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from ood_object_detection_amd import ood
from ood_object_detection_amd.effdet.bench import DetBenchPredict


def main():
    name, size, n_img, bs = (sys.argv[1] if len(sys.argv) > 1 else 'tf_efficientdet_d4'), int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 32, 8
    dev = torch.device('cuda:0')
    model = B.build_model(name, size, 90)
    model.config.soft_nms = True
    model = model.to(dev).to(torch.bfloat16)
    bench = DetBenchPredict(model).to(dev)
    g = torch.Generator(device=dev).manual_seed(7)
    scores = {'in': [], 'ood': []}
    with torch.no_grad():
        for kind in ('in', 'ood'):
            for _ in range(n_img // bs):
                if kind == 'in':
                    low = torch.randn(bs, 3, size // 32, size // 32, device=dev, generator=g)
                    x = torch.nn.functional.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
                else:
                    x = torch.randn(bs, 3, size, size, device=dev, generator=g)
                bench(x.to(torch.bfloat16))
                scores[kind].append(ood.image_scores(bench.last_ood['anchor_energy']).clone())
    s_in, s_ood = torch.cat(scores['in']), torch.cat(scores['ood'])
    print(json.dumps({'config': '%s %dpx bf16 soft-NMS, %d + %d synthetic images (smooth vs white noise), seeded random weights' % (name, size, n_img, n_img),
                      'score': 'max_a(-energy_a)', 'auroc_in_dist_positive': round(ood.auroc(s_in, s_ood), 4),
                      'mean_score_in': round(float(s_in.mean()), 4), 'mean_score_ood': round(float(s_ood.mean()), 4), 'data': 'synthetic'}))


if __name__ == '__main__':
    main()
