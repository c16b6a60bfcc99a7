# owed alpha steps must give bit-identical training to launch-of-its-own alpha steps
import os, sys, subprocess, json
sys.path.insert(0, "/root/repo")
import torch
def run():
    import deep_rl_amd as D
    dev = torch.device("cuda", 0)
    env = D.make("Pendulum-v1", num_envs=64, device=dev, seed=3)
    torch.manual_seed(3)
    actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
    q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
    eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=64, batch_size=256, learning_starts=4)
    eng.reset()
    for it in range(40):
        eng.act()
        if eng.global_step > 6:
            eng.train_step()
    return {"alpha": float(eng.alpha), "log_alpha": float(eng.log_alpha), "steps": eng.alpha_steps, "actor": eng.actor.flat.double().sum().item(),
            "q": eng.q_flat.double().sum().item(), "qt": eng.qt_flat.double().sum().item(), "am": float(eng._alpha_m), "av": float(eng._alpha_v)}
if __name__ == "__main__":
    print(json.dumps(run()))
