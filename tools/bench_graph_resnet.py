"""ResNet-34 headline step with and without Learner.use_graphs() (whole-step hipGraph replay), same process."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

learner, data = bench.build_learner('cuda', 64, 224, 1234)
learner.model.train()
batches = list(data.train_dl)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        loss = learner.train1minibatch(*batches[i % len(batches)], 1e-2, 0.9)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, loss
for _ in range(2):
    run(5)
eager, l0 = run(20)
learner.use_graphs(True)
run(6)
graph, l1 = run(20)
print(json.dumps({'eager_ms': round(eager, 3), 'graph_ms': round(graph, 3), 'loss_eager': l0, 'loss_graph': l1,
                  'graphs': sum(g.graph is not None for g in learner._graphs.values())}))
