"""Empty stand-in so that the unmodified reference sac.py (`import pybullet_envs`, sac.py:5) imports in this container.
TEST INFRASTRUCTURE ONLY.  The Bullet physics is NOT reproduced: oracle/capture_sac_trace.py aliases the script's env id to
Pendulum-v1 (SURVEY.md §8a s8 / BASELINE config 4 re-target the SAC path to Pendulum)."""
