"""Drop-in for the reference's ``train_logReg_swapping`` module (train_logReg_swapping.py:1-141; SURVEY.md §8(f)
ranks 2 and 4): the band-swapped feature dataset (1 real + 8 band-swapped fakes per file) with every transform and
the embedder on the HIP path, and the scikit-learn logistic-regression fit + accuracy + EER on the host, as in the
reference.  Nothing runs at import."""
import os

import numpy as np
import torch

from addvisor_hip import ops as _ops
from audioprocessor import AudioProcessor

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
audio_processor = AudioProcessor()


def find_all_files(metadata):
    """train_logReg_swapping.py:19-27: first CSV field of every line, first 5000."""
    audio_paths = []
    with open(metadata, "r") as f:
        for path in f:
            audio_paths.append(path.strip().split(",")[0])
    return audio_paths[:5000]


def band_swap_features(w_real, w_vocoded):
    """train_logReg_swapping.py:57-92 for one file: time-mean embedder features of the real clip and of its eight
    band-swapped variants -> ``[9, H]`` (row 0 real).  One STFT launch for both clips, one ISTFT launch for the eight
    bands, ONE 9-clip embedder pass."""
    ap = audio_processor
    L = int(ap.audio_length * ap.sampling_rate)
    pair = torch.stack([_fit(w_real, L), _fit(w_vocoded, L)]).to(device, torch.float32)
    X, _, _ = _ops.stft_forward(pair, L, ap.hop_length, ap.win_length, want_mag=False, want_phase=False)
    fakes = _ops.istft_bandswap(X[0:1], X[1:2], L, 0, 64, 8, hop=ap.hop_length, win=ap.win_length)[:, 0]    # [8, L]
    feats = audio_processor.extract_features(torch.cat([pair[0:1], fakes], 0))                              # [9, T, H]
    return feats.mean(dim=1)


def _fit(w, L):
    w = w.reshape(-1)
    return torch.nn.functional.pad(w, (0, L - w.numel())) if w.numel() < L else w[:L]


def generate_time_swap_dataset(metadata, save_dir="time_swap_data", dir_real="LJSpeech_vocoded/",
                               dir_vocoded="LJSpeech_hifigan16K/"):
    """train_logReg_swapping.py:30-102."""
    audio_paths = find_all_files(metadata)
    X, y = [], []
    os.makedirs(save_dir, exist_ok=True)
    for filename in audio_paths:
        w_real, _ = audio_processor.load_audio(os.path.join(dir_real, filename))
        path_vocoded = os.path.join(dir_vocoded, filename + "_vocoded.wav")
        if not os.path.exists(path_vocoded):
            path_vocoded = os.path.join(dir_vocoded, filename)
        w_vocoded, _ = audio_processor.load_audio(path_vocoded)
        with torch.no_grad():
            f = band_swap_features(w_real, w_vocoded).cpu().numpy()
        X.extend(f)
        y.extend([0] + [1] * 8)
    X, y = np.stack(X), np.array(y)
    np.save(os.path.join(save_dir, "X_vocoded_anyband_16k.npy"), X)
    np.save(os.path.join(save_dir, "y_vocoded_anyband_16k.npy"), y)
    return X, y


def equal_error_rate(y_true, y_score):
    """train_logReg_swapping.py:121-122: the ROC point where FPR = 1 - TPR (brentq on the interpolated curve)."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn.metrics import roc_curve
    fpr, tpr, _ = roc_curve(y_true, y_score, pos_label=1)
    return brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0)


def train_logReg_timeswap(X, y, out_path="logReg_ckpts/logReg_vocoded_anyband_16k.joblib"):
    """train_logReg_swapping.py:105-128: stratified 80/20 split (seed 42), ``LogisticRegression(C=1e6, max_iter=10000)``,
    accuracy + EER, ``joblib.dump``.  Returns ``(model, accuracy, eer)``."""
    import joblib
    from sklearn.linear_model import LogisticRegression
    from sklearn.metrics import accuracy_score
    from sklearn.model_selection import train_test_split
    X_train, X_test, y_train, y_test = train_test_split(X, y, test_size=0.2, random_state=42, stratify=y)
    model = LogisticRegression(random_state=42, C=1e6, max_iter=10000)
    model.fit(X_train, y_train)
    acc = accuracy_score(y_test, model.predict(X_test))
    eer = equal_error_rate(y_test, model.predict_proba(X_test)[:, 1])
    print(f"Accuracy: {acc:.4f}")
    print(f"EER: {eer*100:.4f}%")
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        joblib.dump(model, out_path)
    return model, acc, eer


if __name__ == "__main__":
    X_, y_ = generate_time_swap_dataset("metadata/ljspeech_manipulated_metadata.txt", save_dir="time_swap_data")
    train_logReg_timeswap(X_, y_)
